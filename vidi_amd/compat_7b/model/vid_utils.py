from vidi_amd.processors import load_video_7b as load_video, load_audio, process_audio, process_audio_gpu, get_media_length  # noqa: F401
