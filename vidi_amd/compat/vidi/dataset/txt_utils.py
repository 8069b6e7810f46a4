from vidi_amd.processors import tokenizer_image_token, preprocess_chat, chat_template  # noqa: F401
