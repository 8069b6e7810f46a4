from vidi_amd.processors import process_images  # noqa: F401
