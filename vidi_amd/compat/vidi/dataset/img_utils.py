from vidi_amd.processors import process_images, process_images_gpu  # noqa: F401
