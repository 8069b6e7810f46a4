from vidi_amd.model import load_pretrained_model  # noqa: F401
