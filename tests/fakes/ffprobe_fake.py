"""`ffprobe` stand-in (see fake_clip.py): the two duration queries of the reference — Vidi1.5's `get_media_length`
(`-i FILE -show_entries format=duration -v quiet -of csv=p=0`, vid_utils.py:67-80) and Vidi-7B's `get_length`
(`-v error -show_entries format=duration -of default=noprint_wrappers=1:nokey=1 FILE`, inference.py:68-73)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_clip  # noqa: E402


def main(argv):
    a, i, opts, positional = list(argv), 0, {}, []
    while i < len(a):
        if a[i] in ("-i", "-show_entries", "-v", "-of"):
            opts[a[i]] = a[i + 1]; i += 2
        elif a[i].startswith("-"):
            sys.exit(f"ffprobe_fake: unexpected argument {a[i]!r}")
        else:
            positional.append(a[i]); i += 1
    path = opts.get("-i") or (positional[0] if positional else None)
    if path is None or opts.get("-show_entries") != "format=duration" or opts.get("-of") not in ("csv=p=0", "default=noprint_wrappers=1:nokey=1"):
        sys.exit(f"ffprobe_fake: not one of the reference's command lines: {argv}")
    print(f"{fake_clip.read_clip(path)['duration']:.6f}")


if __name__ == "__main__":
    main(sys.argv[1:])
