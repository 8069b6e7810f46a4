"""Checkpoint directory -> state dict -> model: the loading path of `load_pretrained_model(path)` (model/builder.py:24-64) on the
CPU (file handling only); tests/test_gpu_cli.py loads the same directory into the HIP engine and compares logits."""
import json
import os

import pytest
import torch

from vidi_amd import config as C
from vidi_amd.weights import init_random_weights, load_checkpoint, weight_shapes


def write_checkpoint(tmp, cfg, w, towers_inside=True, tower_dirs=True, cfg_tower_keys=True):
    """config.json + model-0000x.safetensors with the reference's parameter names (validated against the reference's own state
    dict in tests/golden/make_golden_dattn.py).  towers_inside=False: the towers live in HF-format tower directories."""
    from safetensors.torch import save_file
    d = cfg.to_dict()
    if not cfg_tower_keys:
        d = {k: v for k, v in d.items() if not k.startswith(("vis_", "aud_"))}
        d["mm_vision_tower"], d["mm_audio_tower"] = "towers/siglip-test", "whisper-test"
    os.makedirs(tmp, exist_ok=True)
    json.dump(d, open(os.path.join(tmp, "config.json"), "w"))
    main = {k: v.contiguous() for k, v in w.items() if towers_inside or not k.startswith(("model.mm_vis.", "model.mm_aud."))}
    keys = sorted(main)
    save_file({k: main[k] for k in keys[: len(keys) // 2]}, os.path.join(tmp, "model-00001-of-00002.safetensors"))
    save_file({k: main[k] for k in keys[len(keys) // 2:]}, os.path.join(tmp, "model-00002-of-00002.safetensors"))
    if not towers_inside and tower_dirs:
        vd, ad = os.path.join(tmp, "towers", "siglip-test"), os.path.join(tmp, "whisper-test")
        os.makedirs(vd); os.makedirs(ad)
        save_file({k[len("model.mm_vis."):]: v.contiguous() for k, v in w.items() if k.startswith("model.mm_vis.")}, os.path.join(vd, "model.safetensors"))
        save_file({"model." + k[len("model.mm_aud."):]: v.contiguous() for k, v in w.items() if k.startswith("model.mm_aud.")},
                  os.path.join(ad, "model.safetensors"))
        json.dump({"vision_config": {"image_size": cfg.vis_image_size, "patch_size": cfg.vis_patch_size, "hidden_size": cfg.vis_hidden_size,
                                     "intermediate_size": cfg.vis_intermediate_size, "num_hidden_layers": cfg.vis_num_layers,
                                     "num_attention_heads": cfg.vis_num_heads, "layer_norm_eps": cfg.vis_ln_eps}}, open(os.path.join(vd, "config.json"), "w"))
        json.dump({"num_mel_bins": cfg.aud_num_mel_bins, "d_model": cfg.aud_d_model, "encoder_layers": cfg.aud_num_layers,
                   "encoder_attention_heads": cfg.aud_num_heads, "encoder_ffn_dim": cfg.aud_ffn_dim,
                   "max_source_positions": cfg.aud_max_source_positions}, open(os.path.join(ad, "config.json"), "w"))


@pytest.mark.parametrize("preset", ["tiny", "tiny_7b"])
def test_checkpoint_round_trip(tmp_path, preset):
    cfg = getattr(C, preset)()
    w = init_random_weights(cfg, seed=4, dtype=torch.float16)
    write_checkpoint(str(tmp_path), cfg, w)
    cfg2 = C.VidiConfig.from_pretrained(str(tmp_path))
    assert cfg2.to_dict() == cfg.to_dict()
    sd = load_checkpoint(str(tmp_path), cfg2)
    assert set(weight_shapes(cfg)) <= set(sd)
    for k in weight_shapes(cfg):
        assert torch.equal(sd[k], w[k]), k


def test_towers_from_their_own_directories(tmp_path):
    """the Vidi checkpoint carries no tower weights and no tower dims (gemma.py:469; multimodal.py:44-57): both come from local
    copies of the tower repositories named in the config"""
    cfg = C.tiny()
    w = init_random_weights(cfg, seed=4, dtype=torch.float16)
    write_checkpoint(str(tmp_path), cfg, w, towers_inside=False, cfg_tower_keys=False)
    cfg2 = C.VidiConfig.from_pretrained(str(tmp_path))
    for k in ("vis_image_size", "vis_hidden_size", "vis_intermediate_size", "vis_num_layers", "vis_num_heads", "aud_num_mel_bins", "aud_d_model",
              "aud_num_layers", "aud_ffn_dim", "aud_max_source_positions", "aud_nb_max_frames"):
        assert getattr(cfg2, k) == getattr(cfg, k), k
    sd = load_checkpoint(str(tmp_path), cfg2)
    for k in weight_shapes(cfg):
        assert torch.equal(sd[k], w[k]), k


def test_missing_towers_raise_a_clear_error(tmp_path):
    cfg = C.tiny()
    w = init_random_weights(cfg, seed=4, dtype=torch.float16)
    write_checkpoint(str(tmp_path), cfg, w, towers_inside=False, tower_dirs=False)
    with pytest.raises(KeyError, match="mm_vision_tower"):
        load_checkpoint(str(tmp_path), cfg)


def test_real_checkpoint_without_tokenizer_is_an_error(tmp_path):
    """no blanket `except`: a checkpoint directory without tokenizer files fails at load time with the tokenizer's own error"""
    from vidi_amd.model import load_pretrained_model
    cfg = C.tiny()
    w = init_random_weights(cfg, seed=4, dtype=torch.float16)
    write_checkpoint(str(tmp_path), cfg, w)
    with pytest.raises(Exception) as e:
        load_pretrained_model(str(tmp_path), engine_factory=lambda c, ww, dt: object())
    assert "okenizer" in str(e.value) or "tokenizer" in str(e.value).lower() or isinstance(e.value, (OSError, ValueError))


def test_checkpoint_from_sharded_torch_pickles(tmp_path):
    """a directory without safetensors: pytorch_model-0000x-of-0000y.bin shards (HF's older export format) load to the same tensors"""
    cfg = C.tiny()
    w = init_random_weights(cfg, seed=4, dtype=torch.float16)
    write_checkpoint(str(tmp_path), cfg, w)
    from vidi_amd.weights import load_safetensors_dir
    sd = load_safetensors_dir(str(tmp_path))
    for f in os.listdir(str(tmp_path)):
        if f.endswith(".safetensors"):
            os.remove(os.path.join(str(tmp_path), f))
    keys = sorted(sd)
    half = len(keys) // 2
    torch.save({k: sd[k] for k in keys[:half]}, os.path.join(str(tmp_path), "pytorch_model-00001-of-00002.bin"))
    torch.save({k: sd[k] for k in keys[half:]}, os.path.join(str(tmp_path), "pytorch_model-00002-of-00002.bin"))
    sd2 = load_checkpoint(str(tmp_path), C.VidiConfig.from_pretrained(str(tmp_path)))
    for k in weight_shapes(cfg):
        assert torch.equal(sd2[k], w[k]), k
