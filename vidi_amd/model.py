"""Reference-compatible model object for the MI355X path.

Mirrors the surface `eval/inference.py` and `model/builder.py` use on the reference's
`DattnGemma2ForCausalLM` (Vidi1.5_9B/vidi/model/lmm/dattn/gemma.py:467-687):

    model.config (mm_splits writable), model.get_model().{text_tokenizer,image_processor,audio_processor},
    model.generation_config.eos_token_id, model.generate(inputs, images=, audios=, audio_sizes=, **kw)
    -> LongTensor[B, n_new] (new tokens only), model.forward(...) -> DattnCausalLMOutputWithPast,
    model.encode_videos(images, audios, audio_sizes), model.prepare_inputs_labels_for_multimodal(...).

The greedy loop is our own: HF GenerationMixin is not involved, so there is no
coupling to a transformers version.  Everything numeric runs in `VidiEngine` (HIP kernels)."""
from __future__ import annotations

import os

from dataclasses import dataclass
from types import SimpleNamespace
from typing import Any, Dict, Optional, Sequence, Tuple

import torch

from .config import VidiConfig
from .engine import MMState, VidiEngine

IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"


@dataclass
class DattnCausalLMOutputWithPast:
    """lmm/dattn/outputs.py:12-20"""
    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Any = None
    past_image_key_values: Any = None
    past_audio_key_values: Any = None
    hidden_states: Any = None
    attentions: Any = None


def strip_image_token(input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                      padding_side: str = "right") -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Integer half of prepare_inputs_labels_for_multimodal (multimodal.py:352-430): drop pads, delete
    the single <image> placeholder (-200), re-pad.  Returns (ids[B,L] with -1 at pads, mask[B,L] bool,
    position_ids[B,L]).  Host-side, bit-exact."""
    ids_cpu = input_ids.detach().cpu()
    am = torch.ones_like(ids_cpu, dtype=torch.bool) if attention_mask is None else attention_mask.detach().cpu().bool()
    rows = []
    for row, m in zip(ids_cpu, am):
        cur = row[m]
        n_img = int((cur == IMAGE_TOKEN_INDEX).sum())
        assert n_img <= 1, "only support at most one image for now."          # multimodal.py:369
        rows.append(cur[cur != IMAGE_TOKEN_INDEX])
    L = max(int(r.shape[0]) for r in rows)
    B = len(rows)
    out = torch.full((B, L), -1, dtype=torch.int64)
    mask = torch.zeros((B, L), dtype=torch.bool)
    pos = torch.zeros((B, L), dtype=torch.int64)
    for i, r in enumerate(rows):
        n = int(r.shape[0])
        if n == 0:
            continue
        if padding_side == "left":
            out[i, -n:] = r; mask[i, -n:] = True; pos[i, -n:] = torch.arange(n)
        else:
            out[i, :n] = r; mask[i, :n] = True; pos[i, :n] = torch.arange(n)
    return out, mask, pos


def strip_image_labels(input_ids: torch.Tensor, labels: torch.Tensor, attention_mask: Optional[torch.Tensor] = None,
                       padding_side: str = "right") -> torch.Tensor:
    """The labels' half of the same function (multimodal.py:358-363, 379-400, 407-437): a row's labels at its attended positions, minus the
    <image> placeholder's, re-padded with IGNORE_INDEX (-100) on the side `strip_image_token` pads."""
    ids_cpu, lab_cpu = input_ids.detach().cpu(), labels.detach().cpu()
    am = torch.ones_like(ids_cpu, dtype=torch.bool) if attention_mask is None else attention_mask.detach().cpu().bool()
    rows = [lab[m][row[m] != IMAGE_TOKEN_INDEX] for row, lab, m in zip(ids_cpu, lab_cpu, am)]
    L = max(int(r.shape[0]) for r in rows)
    out = torch.full((len(rows), L), IGNORE_INDEX, dtype=torch.int64)
    for i, r in enumerate(rows):
        n = int(r.shape[0])
        if n:
            if padding_side == "left":
                out[i, -n:] = r
            else:
                out[i, :n] = r
    return out


def causal_lm_loss(logits: torch.Tensor, labels: torch.Tensor, loss_thres: Optional[float] = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """gemma.py:571-590: fp32 logits flattened to [B * L, V] (that flattened tensor is also what the reference RETURNS as `.logits` when
    labels are given), labels shifted by one with IGNORE_INDEX at the end, mean cross-entropy over the counted positions — or, with
    `config.loss_thres`, over the positions whose loss exceeds it (all of them when none does).  -> (loss, flattened logits)"""
    import torch.nn.functional as F
    flat = logits.float().reshape(-1, logits.shape[-1])
    shift = F.pad(labels.to(flat.device), (0, 1), value=IGNORE_INDEX)[..., 1:].reshape(-1)
    if loss_thres is not None:
        per = F.cross_entropy(flat, shift, ignore_index=IGNORE_INDEX, reduction="none")
        thres = 0.0 if bool(torch.all(per < loss_thres)) else loss_thres
        return torch.mean(per[per > thres]), flat
    return F.cross_entropy(flat, shift, ignore_index=IGNORE_INDEX, reduction="mean"), flat


# HF `GenerationConfig` fields with their neutral values: arguments that alter the decoding and have no counterpart here
_UNSUPPORTED_GENERATION_KWARGS = {
    "min_p": None, "typical_p": 1.0, "epsilon_cutoff": 0.0, "eta_cutoff": 0.0, "top_h": None, "num_beam_groups": 1, "diversity_penalty": 0.0,
    "penalty_alpha": None, "dola_layers": None, "encoder_repetition_penalty": 1.0, "begin_suppress_tokens": None, "forced_bos_token_id": None,
    "forced_eos_token_id": None, "exponential_decay_length_penalty": None, "sequence_bias": None, "renormalize_logits": False,
    "guidance_scale": None, "watermarking_config": None, "prompt_lookup_num_tokens": None, "assistant_model": None, "constraints": None,
    "force_words_ids": None, "prefix_allowed_tokens_fn": None, "remove_invalid_values": False, "token_healing": False, "stop_strings": None,
    "encoder_no_repeat_ngram_size": 0,
}


class _Inner:
    """what `model.get_model()` returns in the reference (DattnGemma2MMModel)"""

    def __init__(self, owner: "VidiForCausalLM"):
        self._owner = owner
        self.text_tokenizer = None
        self.image_processor = None
        self.audio_processor = None


class VidiForCausalLM:
    config_class = VidiConfig

    def __init__(self, config: VidiConfig, weights: Dict[str, torch.Tensor], dtype: torch.dtype = torch.float16,
                 device: str = "cuda", engine=None):
        """`engine`: an object with VidiEngine's interface, built by the caller (the host-logic tests drive this class on the
        CPU that way); by default the HIP engine is constructed here and raises without a HIP device / libvidi_hip.so."""
        self.config = config
        self.dtype = dtype
        self.device = torch.device(device)
        self.engine = engine if engine is not None else VidiEngine(config, weights, dtype=dtype, device=device)
        # HF `GenerationConfig` defaults for the sampling knobs (transformers 4.50 / 4.44: top_k = 50 unless the checkpoint's
        # generation_config.json or the caller says otherwise — the reference forwards **kwargs to HF's generate(), gemma.py:646-655)
        self.generation_config = SimpleNamespace(eos_token_id=config.eos_token_id, pad_token_id=config.pad_token_id,
                                                 temperature=1.0, top_k=50, top_p=1.0)
        self.model = _Inner(self)
        self._mm_cache: Optional[Tuple[Any, MMState]] = None

    # ---- reference accessors ----
    def get_model(self):
        return self.model

    def eval(self):
        return self

    def half(self):
        return self

    def cuda(self):
        return self

    # ---- multimodal encode (multimodal.py:254-265) ----
    def _single(self, xs, name):
        if xs is None:
            return None
        if isinstance(xs, (list, tuple)) or xs.dim() == 5 or (name == "audios" and xs.dim() == 4):
            if len(xs) != 1:
                raise ValueError("internal: per-sample batches are split by _n_videos()/_per_sample before encoding")
            return xs[0]
        return xs

    @staticmethod
    def _n_videos(images, audios) -> int:
        """batch size of the multimodal inputs: images [B,T,3,S,S] / list of [T,3,S,S]; audios [B,C,mel,frames] / list"""
        for xs, nd in ((images, 5), (audios, 4)):
            if xs is None:
                continue
            if isinstance(xs, (list, tuple)):
                return len(xs)
            if xs.dim() == nd:
                return int(xs.shape[0])
        return 1

    @staticmethod
    def _row(xs, i):
        return None if xs is None else [xs[i]]

    def _tower_batch(self, images, audios):
        """ONE pass of each tower over the concatenated frames / 30-s windows of all the batch's videos (multimodal.py:157-169, 216-222:
        `torch.cat` of the samples, then the tower) -> (per-video SigLIP features or None, per-video Whisper features or None).
        Engines without the tower entry points (the CPU test engine) and sharded engines (every rank encodes its own frame range of
        each video) return (None, None): the per-video calls run their towers themselves."""
        eng = self.engine
        B = self._n_videos(images, audios)
        vis = aud = None
        if not hasattr(eng, "siglip_forward") or self._shard_on():
            return vis, aud
        if images is not None and all(int(images[i].shape[0]) > 0 for i in range(B)):
            f = eng.siglip_forward(torch.cat([images[i].to(eng.dev) for i in range(B)], dim=0))
            vis = list(torch.split(f, [int(images[i].shape[0]) for i in range(B)], dim=0))
        if audios is not None and all(int(audios[i].shape[0]) > 0 for i in range(B)):
            f = eng.whisper_forward(torch.cat([audios[i].to(eng.dev) for i in range(B)], dim=0))
            aud = list(torch.split(f, [int(audios[i].shape[0]) for i in range(B)], dim=0))
        return vis, aud

    def _batch_frames(self, images) -> Optional[int]:
        """frames of the WHOLE batch: what the reference's token-budget rule counts (multimodal.py:157-158, 175-180)"""
        if images is None:
            return None
        return int(sum(int(images[i].shape[0]) for i in range(self._n_videos(images, None))))

    def _shard_on(self) -> bool:
        """True once `engine.set_dist()` made this process one rank of a frame-sharded group (either dist mode)"""
        eng = self.engine
        return bool(getattr(eng, "shard_encode", eng.world > 1))

    def _encode_shard(self, img, aud, audio_size, normalizer=None, budget_frames=None):
        """This rank's share of ONE video (every rank is handed all of it): its contiguous range of frames and of 30-s audio windows
        (vidi_amd/shard.py), encoded with the GLOBAL positions (`frame_offset` / `total_frames` for pos_t and the token budget,
        `chunk_offset` + the global `audio_size` for the audio floors) and the WHOLE sample's "any non-zero input" flags
        (`images.sum() != 0`, multimodal.py:203-206, 247-250).  -> (fi, mi, fa, ma) of the local tokens, the shard."""
        from .shard import video_shard
        eng = self.engine
        sh = video_shard(0 if img is None else int(img.shape[0]), 0 if aud is None else int(aud.shape[0]), eng.world, eng.rank)
        fi = mi = fa = ma = None
        kw = {} if normalizer is None else dict(normalizer=normalizer)
        if img is not None:
            kw_i = dict(kw, frame_offset=sh.f0, total_frames=sh.total_frames, sample_flag=eng.sample_flag(img))
            if budget_frames is not None:
                kw_i["budget_frames"] = int(budget_frames)
            fi, mi = eng.encode_video_images(img[sh.f0: sh.f1].to(eng.dev), **kw_i)
        if aud is not None:
            fa, ma = eng.encode_video_audios(aud[sh.c0: sh.c1].to(eng.dev), int(audio_size), chunk_offset=sh.c0, sample_flag=eng.sample_flag(aud), **kw)
        return fi, mi, fa, ma, sh

    def _gather_tokens(self, fi, mi, fa, ma, sh, audio_size, budget_frames=None):
        """The north-star's collective (BASELINE configs[3]): all-gather of the visual (and audio) tokens the ranks encoded, so that every
        rank holds the reference-order rows `encode_videos` returns on one GPU (multimodal.py:254-265; the reference's own sequence-
        parallel path gathers the same way: sequence_parallel/all_to_all.py:361 `Gather.forward`, split.py:73-93 `merge_data`).  The
        row counts of all ranks follow from host integers (shard.gather_counts); ragged shards are padded for the collective and
        narrowed after it.  Per-token math + data movement: bit-identical to the single-rank encode."""
        from .dist import all_gather_rows
        from .shard import gather_counts
        eng = self.engine
        n_img, n_aud = gather_counts(self.config, sh.total_frames if fi is not None else 0, sh.total_windows if fa is not None else 0,
                                     audio_size, eng.world, budget_frames)
        if fi is not None:
            fi, mi = all_gather_rows(fi, n_img, eng.pg), all_gather_rows(mi, n_img, eng.pg)
        if fa is not None:
            fa, ma = all_gather_rows(fa, n_aud, eng.pg), all_gather_rows(ma, n_aud, eng.pg)
        eng.n_token_gathers = getattr(eng, "n_token_gathers", 0) + 2 * (int(fi is not None) + int(fa is not None))
        return fi, mi, fa, ma

    def encode_videos(self, images, audios, audio_sizes):
        """-> (image_features[B,Nv,H], image_mask[B,Nv] bool, audio_features[B,Na,H], audio_mask[B,Na] bool),
        un-normalised like the reference (the normaliser is applied inside the decoder, gemma.py:353-356).  A batch of videos goes
        through each tower in ONE pass and shares the token budget (the reference pools by the batch's total frame count), then
        every video is finished on its own (positions, norms, masks) and the rows are padded like multimodal.py:199, 243.
        Under `engine.set_dist()` (one process per GPU, either dist mode) every rank encodes its frame / window range of each video and
        the tokens are all-gathered: the SAME reference-order tensors on every rank, bit-identical to the single-rank call."""
        eng = self.engine
        B = self._n_videos(images, audios)
        vis, aud = self._tower_batch(images, audios) if B > 1 else (None, None)
        budget = self._batch_frames(images) if B > 1 else None
        per = []
        for i in range(B):
            img = self._single(self._row(images, i) if B > 1 else images, "images")
            au = self._single(self._row(audios, i) if B > 1 else audios, "audios")
            fi = mi = fa = ma = None
            if self._shard_on():
                asz = None if au is None else int(audio_sizes[i])
                fi, mi, fa, ma, sh = self._encode_shard(img, au, asz, budget_frames=budget)
                fi, mi, fa, ma = self._gather_tokens(fi, mi, fa, ma, sh, asz, budget_frames=budget)
                per.append((fi, None if mi is None else mi.bool(), fa, None if ma is None else ma.bool()))
                continue
            if img is not None:
                kw = {} if budget is None else dict(budget_frames=budget)
                if vis is not None:
                    kw["vis_features"] = vis[i]
                fi, mi = eng.encode_video_images(img.to(eng.dev), **kw)
            if au is not None:
                kw = {} if aud is None else dict(aud_features=aud[i])
                fa, ma = eng.encode_video_audios(au.to(eng.dev), int(audio_sizes[i]), **kw)
            per.append((fi, None if mi is None else mi.bool(), fa, None if ma is None else ma.bool()))
        pad = torch.nn.utils.rnn.pad_sequence
        cat = lambda k: None if per[0][k] is None else pad([p[k] for p in per], batch_first=True)     # noqa: E731
        return cat(0), cat(1), cat(2), cat(3)

    def encode_mm_state(self, images, audios, audio_sizes, vis_features=None, aud_features=None, budget_frames=None) -> MMState:
        """encode + run the query-independent multimodal stream through all layers (caches) for ONE video.

        Under `engine.set_dist(...)` (one process per GPU) every rank is handed the SAME video and keeps only its share: a
        contiguous range of frames and of 30-s audio windows (vidi_amd/shard.py), encoded with the global positions, streamed
        through the 42 layers locally (the diagonal stream never mixes tokens) and left resident as this rank's K/V shard.
        The flags the reference derives from the whole sample (`images.sum() != 0`, multimodal.py:203-206, 247-250) are taken
        from the whole sample here too.  `vis_features` / `aud_features`: this video's tower outputs when a batch of videos went
        through the towers together; `budget_frames`: the batch's total frames for the token-budget rule (see encode_videos)."""
        eng = self.engine
        img = self._single(images, "images")
        aud = self._single(audios, "audios")
        fi = mi = fa = ma = None
        nz = eng.normalizer
        if self._shard_on():
            asz = None if aud is None else int(audio_sizes[0])
            fi, mi, fa, ma, sh = self._encode_shard(img, aud, asz, normalizer=nz, budget_frames=budget_frames)
            if getattr(eng, "dist_mode", "sharded_stream") == "gather_tokens":
                # the all-gather of visual / audio tokens; from here on the decoder is replicated (stream, caches, cross-attention of one GPU)
                fi, mi, fa, ma = self._gather_tokens(fi, mi, fa, ma, sh, asz, budget_frames=budget_frames)
        else:
            kw_i, kw_a = {}, {}
            if budget_frames is not None:
                kw_i["budget_frames"] = int(budget_frames)
            if vis_features is not None:
                kw_i["vis_features"] = vis_features
            if aud_features is not None:
                kw_a["aud_features"] = aud_features
            if img is not None:
                fi, mi = eng.encode_video_images(img.to(eng.dev), normalizer=nz, **kw_i)
            if aud is not None:
                fa, ma = eng.encode_video_audios(aud.to(eng.dev), int(audio_sizes[0]), normalizer=nz, **kw_a)
        st = eng.mm_stream_prefill(fi, mi, fa, ma, pre_normalized=True)
        st.image_attention_mask, st.audio_attention_mask = mi, ma
        return st

    # ---- process-group helpers of the sharded path ----
    def _backend(self) -> str:
        import torch.distributed as dist
        return dist.get_backend(self.engine.pg)

    def _bcast0(self, t: torch.Tensor) -> torch.Tensor:
        """rank 0's value of a small tensor on every rank of the engine's group"""
        from .dist import broadcast0
        return broadcast0(t, self.engine.pg)

    def _stack_rows(self, rows, pad, kwargs):
        """answers of a batch decoded row by row ([k, n_i] each; k = num_return_sequences under beam search, else 1) -> [sum k, max n_i],
        shorter ones padded on the right as HF pads finished rows; under `return_dict_in_generate` (beam search) the same object again"""
        as_dict = not torch.is_tensor(rows[0])
        seqs = [r.sequences if as_dict else r for r in rows]
        fill = int(pad)
        n = max(int(q.shape[1]) for q in seqs)
        out = torch.full((sum(int(q.shape[0]) for q in seqs), n), fill, dtype=torch.int64, device=self.engine.dev)
        at = 0
        for q in seqs:
            out[at: at + q.shape[0], : q.shape[1]] = q
            at += q.shape[0]
        if not as_dict:
            return out
        from types import SimpleNamespace
        if hasattr(rows[0], "sequences_scores"):                                  # beam search
            sc = [r.sequences_scores for r in rows]
            return SimpleNamespace(sequences=out, sequences_scores=None if sc[0] is None else torch.cat(sc))

        def steps(name):                                                          # per-step [B, V] tensors; rows that stopped earlier: -inf / 0 rows are not invented
            per = [getattr(r, name) for r in rows]
            if per[0] is None:
                return None
            if len({len(p) for p in per}) != 1:
                raise NotImplementedError(f"`{name}` of a batch decoded row by row whose rows stop at different lengths")
            return tuple(torch.cat([p[i] for p in per], dim=0) for i in range(len(per[0])))
        return SimpleNamespace(sequences=out, scores=steps("scores"), logits=steps("logits"), past_key_values=None)

    # ---- text prefill + greedy decode ----
    def _prefill(self, ids: torch.Tensor, mask: torch.Tensor, pos: torch.Tensor, mm: Optional[MMState], max_new: int):
        eng = self.engine
        B, L = ids.shape
        ts = eng.new_text_state(B, L + max_new + 1)
        emb = eng.embed_tokens(ids.to(eng.dev))                         # pads (-1) -> zero rows (multimodal.py:423-426)
        hn = eng.text_forward(emb, pos.reshape(-1).to(eng.dev), ts, mm, Lq=L, new_mask=mask.to(eng.dev))
        lens = mask.sum(-1).to(eng.dev)
        ts.n_valid = lens.clone()
        # logits of each row's LAST VALID token.  (HF + right padding would read position -1, i.e. a
        # pad slot for shorter rows; the reference CLI never batches, see DESIGN.md "batch semantics".)
        last = hn.view(B, L, -1)[torch.arange(B, device=eng.dev), lens - 1]
        return ts, last

    @torch.no_grad()
    def generate(self, inputs: Optional[torch.Tensor] = None, images=None, image_sizes=None, audios=None,
                 audio_sizes: Optional[Sequence[int]] = None, mm_state: Optional[MMState] = None, **kwargs) -> torch.Tensor:
        """gemma.py:603-655.  Accepts do_sample/max_new_tokens/use_cache/disable_compile/pad_token_id/
        attention_mask/position_ids, for do_sample=True: temperature/top_k/top_p/generator (HF warper semantics,
        vidi_amd/sampling.py), and num_beams/length_penalty/early_stopping/num_return_sequences (HF beam search, vidi_amd/beam.py);
        the reference CLI uses do_sample=False, one beam."""
        if "inputs_embeds" in kwargs:
            raise NotImplementedError("`inputs_embeds` is not supported")            # gemma.py:615-616
        # HF generation arguments that WOULD change the output and that this loop does not implement: refuse them instead of ignoring them
        # (anything a caller needs from this list can be handed over as a ready `logits_processor` / `stopping_criteria` object)
        for k in _UNSUPPORTED_GENERATION_KWARGS:
            v = kwargs.get(k, None)
            if v is not None and v is not False and v != _UNSUPPORTED_GENERATION_KWARGS[k]:
                raise NotImplementedError(f"generate(): `{k}` is not implemented by the MI355X path (pass an equivalent `logits_processor` / "
                                          "`stopping_criteria` object, or use the arguments listed in INTEGRATION.md section 5)")
        do_sample = bool(kwargs.get("do_sample", False))
        num_beams = int(kwargs.get("num_beams", None) or 1)
        n_ret = int(kwargs.get("num_return_sequences", None) or 1)
        if num_beams == 1 and n_ret > 1 and not do_sample:
            raise ValueError("Greedy methods (do_sample != True) without beam search do not support `num_return_sequences` different than 1 "
                             f"(got {n_ret}).")                                    # HF's own rule
        if num_beams > 1 and kwargs.get("streamer") is not None:
            raise ValueError("`streamer` cannot be used with beam search (yet!). Make sure that `num_beams` is set to 1.")   # HF's own rule
        # HF resolves the generation lengths ONCE, against the whole batch's padded prompt (`inputs_embeds.shape[1]`); the row-by-row paths of
        # `_generate_resolved` recurse into it (not into this function) with the resolved values, so nothing is subtracted twice and no caller-visible
        # keyword can switch the resolution off
        max_new = kwargs.get("max_new_tokens", None)
        if max_new is None:
            # HF's `_prepare_generated_length` under `inputs_embeds` (how the reference drives it, gemma.py:646-655): `max_length` (default
            # 20) counts the batch's embedded prompt positions only when the caller set it; `max_new_tokens` wins when both are given
            max_length = kwargs.get("max_length", None)
            n_prompt = int(strip_image_token(inputs, kwargs.get("attention_mask", None))[0].shape[1])
            max_new = 20 if max_length is None else int(max_length) - n_prompt
            if max_new <= 0:
                raise ValueError(f"Input length of input_ids is {n_prompt}, but `max_length` is set to {max_length}. This can lead to "
                                 "unexpected behavior. You should consider increasing `max_length` or, better yet, setting `max_new_tokens`.")
        kwargs = dict(kwargs, max_new_tokens=int(max_new))
        if kwargs.get("min_length"):
            # `min_length` counts the embedded prompt too (HF's `_prepare_generated_length` under `inputs_embeds`): what is left applies to the new tokens
            n_prompt = int(strip_image_token(inputs, kwargs.get("attention_mask", None))[0].shape[1])
            kwargs["min_length"] = max(int(kwargs["min_length"]) - n_prompt, 0)
        return self._generate_resolved(inputs, images, audios, audio_sizes, mm_state, kwargs)

    def _generate_resolved(self, inputs, images, audios, audio_sizes, mm_state, kwargs):
        """generate() behind its argument checks, with `max_new_tokens` / `min_length` already counted in NEW tokens (resolved once per call)"""
        do_sample = bool(kwargs.get("do_sample", False))
        num_beams = int(kwargs.get("num_beams", None) or 1)
        n_ret = int(kwargs.get("num_return_sequences", None) or 1)
        max_new = int(kwargs["max_new_tokens"])
        kwargs = dict(kwargs)
        eos = kwargs.get("eos_token_id", self.generation_config.eos_token_id)
        eos_list = [int(e) for e in (eos if isinstance(eos, (list, tuple)) else [eos]) if e is not None]     # HF allows a list ([1, 107] for Gemma2)
        pad = kwargs.get("pad_token_id", None)
        pad = (eos_list[0] if eos_list else 0) if pad is None else pad
        if num_beams > 1:
            pad = pad or (eos_list[0] if eos_list else -1)                       # HF's beam search fills with `pad_token_id or eos_token_id[0]`
        attention_mask = kwargs.get("attention_mask", None)
        eng = self.engine
        if eng.mistral and attention_mask is not None and int(attention_mask[:, -1].sum()) != attention_mask.shape[0]:
            # Vidi_7B/.../mistral.py:366-373: batched generation with right padding is rejected
            raise ValueError("You are attempting to perform batched generation with padding_side='right' this may lead to "
                             "unexpected behaviour for Flash Attention version of Mistral. Make sure to call "
                             "`tokenizer.padding_side  = 'left'` before tokenizing the input. ")
        if mm_state is None and self._n_videos(images, audios) > 1:
            # one video PER ROW (multimodal.py:156-252 batches them with pad_sequence + masks): rows never interact, so each
            # row is answered against its own video, unpadded, and the answers are re-padded — the same tokens
            B = self._n_videos(images, audios)
            assert inputs.shape[0] == B, "one prompt per video"
            # both towers ONCE over the concatenated frames / windows of the batch, and the batch's total frame count for the token
            # budget (multimodal.py:157-180); from there on the videos are independent
            vis, aud = self._tower_batch(images, audios)
            budget = self._batch_frames(images)
            rows = []
            for i in range(B):
                kw = dict(kwargs)
                kw["attention_mask"] = None
                am_i = None if attention_mask is None else attention_mask[i].bool().cpu()
                ids_i = inputs[i].cpu() if am_i is None else inputs[i].cpu()[am_i]
                mm_i = self.encode_mm_state(self._row(images, i), self._row(audios, i), None if audio_sizes is None else [audio_sizes[i]],
                                            vis_features=None if vis is None else vis[i], aud_features=None if aud is None else aud[i],
                                            budget_frames=budget)
                rows.append(self._generate_resolved(ids_i[None], None, None, None, mm_i, kw))
                del mm_i
            return self._stack_rows(rows, pad, kwargs)
        if mm_state is None and (images is not None or audios is not None):
            mm_state = self.encode_mm_state(images, audios, audio_sizes)
        if attention_mask is not None and inputs.shape[0] > 1 and not bool(attention_mask[:, 0].bool().all()):
            # LEFT-padded batch (what Vidi-7B demands for batched generation, mistral.py:366-373; multimodal.py:413-421 re-pads
            # on the tokenizer's side): rows never interact, so every row is decoded unpadded against the shared video state
            rows = []
            for i in range(inputs.shape[0]):
                kw = dict(kwargs)
                kw["attention_mask"] = None
                rows.append(self._generate_resolved(inputs[i].cpu()[attention_mask[i].bool().cpu()][None], None, None, None, mm_state, kw))
            return self._stack_rows(rows, pad, kwargs)
        ids, mask, pos = strip_image_token(inputs, attention_mask)
        if num_beams > 1:
            return self._generate_beams(ids, mask, pos, mm_state, max_new, num_beams, eos_list, kwargs)
        if n_ret > 1:
            # sampling with several answers per prompt: HF expands every row `num_return_sequences` times before the prefill
            # (`_expand_inputs_for_generation`) and draws all rows of a step in one multinomial call — same here, same draws
            ids, mask, pos = (t.repeat_interleave(n_ret, dim=0) for t in (ids, mask, pos))
        ts, last = self._prefill(ids, mask, pos, mm_state, max_new)
        B = ids.shape[0]
        out = torch.full((B, max_new), int(pad), dtype=torch.int64, device=eng.dev)
        finished = torch.zeros(B, dtype=torch.bool, device=eng.dev)
        eos_t = torch.tensor(eos_list, dtype=torch.int64, device=eng.dev)
        # HF generation hooks (GenerationMixin semantics).  The reference drives HF's loop with `inputs_embeds` (gemma.py:646-655), so the
        # `input_ids` these callables see hold the NEW tokens only ([B, 0] at the first step):
        #   logits_processor  [callable(input_ids, scores) -> scores]   applied to the fp32 scores before the argmax / the sampling warpers
        #   stopping_criteria [callable(input_ids, scores) -> bool | BoolTensor[B]]   rows it flags finish like rows that emitted EOS
        #   streamer          .put(LongTensor[B]) per step, .end() after the last one
        # plain keyword arguments HF turns into processors / criteria itself (repetition_penalty, no_repeat_ngram_size, bad_words_ids,
        # min_length, min_new_tokens, suppress_tokens, max_time) come first, in HF's order; the caller's own lists follow
        from .sampling import generation_kwargs_processors
        kw_procs, kw_crits = generation_kwargs_processors(kwargs, eos_list, eng.dev)
        processors = kw_procs + list(kwargs.get("logits_processor") or [])
        criteria = kw_crits + list(kwargs.get("stopping_criteria") or [])
        streamer = kwargs.get("streamer")

        # `return_dict_in_generate=True`: HF's GenerateDecoderOnlyOutput fields that have a meaning here — `.sequences`, and per step
        # `.scores` (output_scores: the fp32 scores after the processors and, when sampling, the warpers) / `.logits` (output_logits: raw)
        as_dict = bool(kwargs.get("return_dict_in_generate"))
        keep_scores = [] if as_dict and kwargs.get("output_scores") else None
        keep_logits = [] if as_dict and kwargs.get("output_logits") else None

        def pick(h, step):
            logits, idx = eng.logits_argmax(h)                          # lm_head + final softcap (in place) + argmax
            if keep_logits is not None:
                keep_logits.append(logits.float().clone())
            if processors:
                logits = logits.float()
                for proc in processors:
                    logits = proc(out[:, :step], logits)
                idx = torch.argmax(logits, dim=-1)
            if do_sample:
                from .sampling import sample, warp_logits
                gc = self.generation_config
                knob = lambda k: kwargs[k] if k in kwargs else getattr(gc, k, None)     # noqa: E731  an explicit None switches a warper off
                logits = warp_logits(logits, knob("temperature"), knob("top_k"), knob("top_p"))
                idx = sample(logits, kwargs.get("generator"))
            if keep_scores is not None:
                keep_scores.append(logits.float().clone())
            return idx, logits

        # sharded + sampling: the replicated text streams must draw the SAME token on every rank (each rank has its own RNG state, and
        # a divergent token or stop decision would mix partials of different queries in the next all-gather or strand a rank in it)
        sync_pick = do_sample and eng.world > 1
        nxt, scores = pick(last, 0)
        if sync_pick:
            nxt = self._bcast0(nxt)
        n_done = 0
        # VIDI_DECODE_GRAPH=1: decode steps are replayed from a hipGraph (device-side cache position, no per-launch
        # host work).  Measured on MI355X (60-min video): replay 19.3 ms/token vs 21.0 eager, capture 126 ms —
        # it only pays for generations of ~80+ tokens, so it is opt-in; the sharded path stays eager.
        use_graph = (not do_sample and not processors and not criteria and not as_dict and max_new >= int(os.environ.get("VIDI_DECODE_GRAPH_MIN", "8"))
                     and os.environ.get("VIDI_DECODE_GRAPH", "0") == "1" and (eng.world == 1 or self._backend() == "nccl"))
        replay = None
        if streamer is not None:
            # HF generate() hands the streamer the prompt ids first — [B, 0] when driven by inputs_embeds, as the reference is
            # (gemma.py:646-655) — then every new token: TextStreamer(skip_prompt=True) drops exactly that first call
            streamer.put(out[:, :0].cpu())
        for step in range(max_new):
            nxt = torch.where(finished, torch.full_like(nxt, int(pad)), nxt)
            out[:, step] = nxt
            n_done = step + 1
            if streamer is not None:
                streamer.put(nxt.cpu())
            finished = finished | torch.isin(nxt, eos_t)
            for crit in criteria:
                stop = crit(out[:, :n_done], scores)
                finished = finished | (stop.to(finished.device).bool() if torch.is_tensor(stop) else torch.full_like(finished, bool(stop)))
            if criteria and eng.world > 1:
                # user criteria may be non-deterministic (MaxTime): every rank follows rank 0's decision, or a rank that stops alone
                # strands the others in the next layer's all-gather
                finished = self._bcast0(finished.to(torch.uint8)).bool()
            if bool(finished.all()) or step == max_new - 1:          # one D2H sync per token, like HF's stopping criteria
                break
            if use_graph:
                if replay is None:
                    nxt, replay = eng.make_decode_graph(ts, mm_state, nxt)
                else:
                    nxt = replay(nxt).clone()
                continue
            emb = eng.embed_tokens(nxt)
            posn = ts.n_valid.clone()                                  # HF: position = cumsum(mask) - 1 of the new token
            ts.n_valid += 1
            hn = eng.text_forward(emb, posn, ts, mm_state, Lq=1)
            nxt, scores = pick(hn, step + 1)
            if sync_pick:
                nxt = self._bcast0(nxt)
        if streamer is not None:
            streamer.end()
        if as_dict:
            from types import SimpleNamespace
            # (a step's scores exist only if the loop went on to that step: n_done entries, like HF's tuples)
            return SimpleNamespace(sequences=out[:, :n_done], scores=None if keep_scores is None else tuple(keep_scores[:n_done]),
                                   logits=None if keep_logits is None else tuple(keep_logits[:n_done]), past_key_values=None)
        return out[:, :n_done]

    def _generate_beams(self, ids, mask, pos, mm_state, max_new, num_beams, eos_list, kwargs):
        """`num_beams > 1` (HF `GenerationMixin._beam_search`, reached through gemma.py:646-655's `**kwargs`): vidi_amd/beam.py holds the
        search; here: the rows expanded to `num_beams` each before the text prefill, one decode step per search step with the text K/V
        rows re-gathered from their parent beams, and HF's output conventions (the new tokens only, as under `inputs_embeds`; unfinished
        positions hold `pad_token_id`, or the first EOS id when that is None / 0 — HF's `pad_token_id or eos_token_id[0]`;
        `return_dict_in_generate=True` -> an object with `.sequences` and, with `output_scores=True`, `.sequences_scores`)."""
        from types import SimpleNamespace
        from .beam import beam_search
        from .sampling import generation_kwargs_processors
        eng = self.engine
        B, nb = ids.shape[0], num_beams
        nrs = int(kwargs.get("num_return_sequences", None) or 1)
        if nrs > nb:
            raise ValueError(f"`num_return_sequences` ({nrs}) has to be smaller or equal to `num_beams` ({nb}).")
        rep = lambda t: t.repeat_interleave(nb, dim=0)                          # noqa: E731  rows b*nb .. b*nb + nb - 1 = the beams of prompt b
        ts, last = self._prefill(rep(ids), rep(mask), rep(pos), mm_state, max_new)
        kw_procs, kw_crits = generation_kwargs_processors(kwargs, eos_list, eng.dev)
        processors = kw_procs + list(kwargs.get("logits_processor") or [])
        criteria = kw_crits + list(kwargs.get("stopping_criteria") or [])
        if eng.world > 1 and criteria:
            # the greedy loop broadcasts rank 0's stop decisions; beam search evaluates its criteria inside vidi_amd/beam.py on every rank, and a
            # wall-clock (`max_time`) or caller-supplied criterion may differ between ranks: one rank would leave the per-layer all-gathers early
            raise NotImplementedError("beam search over a sharded video takes no `max_time` / `stopping_criteria` (the ranks could stop at different steps)")
        do_sample = bool(kwargs.get("do_sample", False))
        if do_sample:
            # beam-search multinomial sampling: HF appends the warpers to the processors (they see the log-probabilities)
            from .sampling import warp_logits
            gc = self.generation_config
            knob = lambda k: kwargs[k] if k in kwargs else getattr(gc, k, None)     # noqa: E731
            keep = max(2, len(eos_list) + 1)                               # HF: top-k / top-p keep at least n_eos + 1 tokens under beams
            processors = processors + [lambda input_ids, scores: warp_logits(scores, knob("temperature"), knob("top_k"), knob("top_p"), keep)]
            if eng.world > 1:
                raise NotImplementedError("beam-search sampling over a sharded video (every rank would draw its own continuations)")
        fill = kwargs.get("pad_token_id", None) or (eos_list[0] if eos_list else -1)
        length_penalty = kwargs.get("length_penalty", None)

        def step_logits(tokens, parents):
            if parents is not None:
                eng.reorder_text_state(ts, parents)
            emb = eng.embed_tokens(tokens)
            posn = ts.n_valid.clone()
            ts.n_valid += 1
            return eng.logits_argmax(eng.text_forward(emb, posn, ts, mm_state, Lq=1))[0]

        seqs, scores = beam_search(step_logits, eng.logits_argmax(last)[0], B, nb, int(self.config.vocab_size), max_new, eos_list, int(fill),
                                   processors, criteria, 1.0 if length_penalty is None else float(length_penalty),
                                   kwargs.get("early_stopping", False) or False, nrs, do_sample=do_sample, generator=kwargs.get("generator"))
        if kwargs.get("return_dict_in_generate"):
            return SimpleNamespace(sequences=seqs, sequences_scores=scores if kwargs.get("output_scores") else None)
        return seqs

    # ---- forward (gemma.py:484-601): prefill-style call returning logits ----
    @torch.no_grad()
    def forward(self, input_ids: Optional[torch.Tensor] = None, attention_mask: Optional[torch.Tensor] = None,
                images=None, audios=None, audio_sizes=None, mm_state: Optional[MMState] = None,
                logits_to_keep: int = 0, labels: Optional[torch.Tensor] = None, **kwargs) -> DattnCausalLMOutputWithPast:
        eng = self.engine
        if mm_state is None and self._n_videos(images, audios) > 1:
            # one video per row: rows are independent -> run them one by one, unpadded, and lay the logits back out in the
            # padded [B, L, V] frame of the reference (pad slots hold zeros there, unspecified values in the reference)
            B = self._n_videos(images, audios)
            ids_all, mask_all, _ = strip_image_token(input_ids, attention_mask)
            L = ids_all.shape[1]
            outs, states = [], []
            vis, aud = self._tower_batch(images, audios)                       # the towers once over the batch; batch-wide token budget
            budget = self._batch_frames(images)
            for i in range(B):
                am_i = None if attention_mask is None else attention_mask[i].bool().cpu()
                ids_i = input_ids[i].cpu() if am_i is None else input_ids[i].cpu()[am_i]
                mm_i = self.encode_mm_state(self._row(images, i), self._row(audios, i), None if audio_sizes is None else [audio_sizes[i]],
                                            vis_features=None if vis is None else vis[i], aud_features=None if aud is None else aud[i],
                                            budget_frames=budget)
                r = self.forward(ids_i[None], mm_state=mm_i, logits_to_keep=0)
                outs.append(r.logits[0])
                states.append((r.past_key_values, r.past_image_key_values))
            full = torch.zeros((B, L, outs[0].shape[-1]), dtype=outs[0].dtype, device=eng.dev)
            for i, o in enumerate(outs):
                full[i, : o.shape[0]] = o                                         # right padding (multimodal.py:423-432)
            keep = full if logits_to_keep == 0 else full[:, -logits_to_keep:]
            loss = None
            if labels is not None:
                loss, keep = causal_lm_loss(keep, strip_image_labels(input_ids, labels, attention_mask), getattr(self.config, "loss_thres", None))
            return DattnCausalLMOutputWithPast(loss=loss, logits=keep, past_key_values=[s[0] for s in states],
                                               past_image_key_values=[s[1] for s in states],
                                               past_audio_key_values=[s[1] for s in states])
        ids, mask, pos = strip_image_token(input_ids, attention_mask)
        if mm_state is None and (images is not None or audios is not None):
            mm_state = self.encode_mm_state(images, audios, audio_sizes)
        B, L = ids.shape
        ts = eng.new_text_state(B, L + 1)
        emb = eng.embed_tokens(ids.to(eng.dev))
        hn = eng.text_forward(emb, pos.reshape(-1).to(eng.dev), ts, mm_state, Lq=L, new_mask=mask.to(eng.dev))
        hn = hn.view(B, L, -1)
        keep = hn if logits_to_keep == 0 else hn[:, -logits_to_keep:]
        Bk, Lk, H = keep.shape
        logits = eng.logits_argmax(keep.reshape(Bk * Lk, H).contiguous())[0].view(Bk, Lk, -1)      # lm_head + final softcap (gemma.py:562-569)
        loss = None
        if labels is not None:
            loss, logits = causal_lm_loss(logits, strip_image_labels(input_ids, labels, attention_mask), getattr(self.config, "loss_thres", None))
        if eng.mistral:
            # Vidi-7B (mistral.py:586-616): the same mean cross-entropy (over the positions that are not IGNORE_INDEX) but NO logits when
            # labels are given, and fp32 logits otherwise
            logits = None if labels is not None else logits.float()
        return DattnCausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=ts,
                                           past_image_key_values=mm_state, past_audio_key_values=mm_state)

    __call__ = forward

    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, image_sizes, audios, audio_sizes):
        """multimodal.py:339-451 signature; returns the same 10-tuple (labels re-aligned with the embedded positions, IGNORE_INDEX at pads)."""
        if input_ids.shape[1] == 1:
            return input_ids, position_ids, attention_mask, past_key_values, None, labels, None, None
        ids, mask, pos = strip_image_token(input_ids, attention_mask)
        eng = self.engine
        # raw embed_tokens output like the reference (multimodal.py:372-432): the sqrt(H) normalizer is the decoder's (gemma.py:353-356)
        emb = eng.embed_tokens(ids.to(eng.dev), normalize=False).view(ids.shape[0], ids.shape[1], -1)
        fi, mi, fa, ma = self.encode_videos(images, audios, audio_sizes)
        new_labels = None if labels is None else strip_image_labels(input_ids, labels, attention_mask).to(labels.device)
        return (None, pos if position_ids is not None else None, mask if attention_mask is not None else None,
                past_key_values, emb, new_labels, fi, mi, fa, ma)


def load_pretrained_model(model_name_or_path: str, load_8bit: bool = False, load_4bit: bool = False, device_map: str = "auto",
                          device: str = "cuda", use_flash_attn: bool = True, **kwargs):
    """model/builder.py:24-64 signature.  Loads config.json + *.safetensors from a local directory
    (fp16 like the reference: builder.py:41), or builds a synthetic model when `kwargs['synthetic']`
    names a preset.  Returns (model, tokenizer, image_processor, audio_processor)."""
    if load_8bit or load_4bit:
        raise NotImplementedError("bitsandbytes quantised loading is out of scope")
    from .weights import init_random_weights, load_checkpoint
    dtype = kwargs.pop("torch_dtype", torch.float16)
    synthetic = kwargs.pop("synthetic", None)
    if synthetic is not None:
        from . import config as C
        cfg = getattr(C, synthetic)() if isinstance(synthetic, str) else synthetic
        weights = init_random_weights(cfg, seed=int(kwargs.pop("seed", 3)), dtype=dtype, device=device)
    else:
        cfg = VidiConfig.from_pretrained(model_name_or_path)
        weights = load_checkpoint(model_name_or_path, cfg)
    engine_factory = kwargs.pop("engine_factory", None)
    engine = engine_factory(cfg, weights, dtype) if engine_factory is not None else None
    model = VidiForCausalLM(cfg, weights, dtype=dtype, device=device, engine=engine)
    tok = img_proc = aud_proc = None
    from .processors import build_processors
    if synthetic is None:
        tok, img_proc, aud_proc = build_processors(model_name_or_path, cfg)        # a real checkpoint without its tokenizer / processor files is an error
    else:
        try:
            tok, img_proc, aud_proc = build_processors(model_name_or_path, cfg)
        except Exception as e:              # synthetic runs usually have no tokenizer files: processors stay None, say so
            import warnings
            warnings.warn(f"synthetic model without tokenizer/processor files ({type(e).__name__}: {e}); processors are None")
    model.get_model().text_tokenizer, model.get_model().image_processor, model.get_model().audio_processor = tok, img_proc, aud_proc
    model.generation_config.eos_token_id = cfg.eos_token_id
    gc_path = os.path.join(str(model_name_or_path), "generation_config.json")
    if os.path.isfile(gc_path):                                          # the checkpoint's own sampling defaults, as HF's from_pretrained reads them
        import json
        with open(gc_path) as f:
            gc = json.load(f)
        for k in ("temperature", "top_k", "top_p", "pad_token_id"):
            if gc.get(k) is not None:
                setattr(model.generation_config, k, gc[k])
    return model, tok, img_proc, aud_proc
