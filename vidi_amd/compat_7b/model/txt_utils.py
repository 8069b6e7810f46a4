from vidi_amd.processors import tokenizer_image_token, chat_template  # noqa: F401
from vidi_amd.processors import preprocess_chat_mistral as preprocess_chat  # noqa: F401
