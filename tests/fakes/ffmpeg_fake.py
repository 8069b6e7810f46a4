"""`ffmpeg` stand-in (see fake_clip.py): understands exactly the command line the reference builds (vid_utils.py:24-49, whisper's audio
loading recipe) and refuses anything else, so a changed command line fails the test instead of being silently accepted."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import fake_clip  # noqa: E402


def main(argv):
    a = list(argv)
    opts, i = {}, 0
    if a[-1] != "-":
        sys.exit("ffmpeg_fake: output must be '-' (stdout)")
    a = a[:-1]
    while i < len(a):
        k = a[i]
        if k == "-nostdin":
            opts[k] = True; i += 1
        elif k in ("-threads", "-i", "-ss", "-t", "-f", "-ac", "-acodec", "-ar"):
            opts[k] = a[i + 1]; i += 2
        else:
            sys.exit(f"ffmpeg_fake: unexpected argument {k!r}")
    if not (opts.get("-nostdin") and opts.get("-f") == "s16le" and opts.get("-ac") == "1" and opts.get("-acodec") == "pcm_s16le" and "-i" in opts and "-ar" in opts):
        sys.exit(f"ffmpeg_fake: not the reference's command line: {argv}")
    meta = fake_clip.read_clip(opts["-i"])
    start = float(opts.get("-ss", 0.0))
    dur = float(opts["-t"]) if "-t" in opts else None
    sys.stdout.buffer.write(fake_clip.pcm(meta, int(opts["-ar"]), start, dur).tobytes())


if __name__ == "__main__":
    main(sys.argv[1:])
