"""Python handles on the native engines of libwmar_hip.so (GPT decode loop, VQGAN)."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from .. import _lib
from ..utils.synth import GPTConfig, VQConfig


def _require_cuda(t: torch.Tensor, what: str):
    if not t.is_cuda:
        raise RuntimeError(f"wmar_amd: {what} must live on the MI355X (no CPU implementation)")


class GPTEngine:
    """minGPT with a static KV cache; replaces GPT.forward_with_past + sample_with_past
    (deps/taming/modules/transformer/mingpt.py:183-214, 326-368)."""

    def __init__(self, cfg: GPTConfig, state: Dict[str, torch.Tensor], max_batch: int = 64, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        L = _lib.load()
        tensors = {}
        for k, v in state.items():
            if k.endswith("attn.mask"):
                continue
            t = v.detach().to(device=self.device, dtype=torch.float32).contiguous()
            tensors[k] = t
        names, ptrs, n = _lib.tensor_table(tensors)
        c = _lib.GptConfig(cfg.vocab_size, cfg.block_size, cfg.n_layer, cfg.n_head, cfg.n_embd, self.max_batch)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.wmar_gpt_create(C.byref(c), names, ptrs, n, _lib.stream_ptr(self.device), C.byref(h)))
        self._h = h
        self._L = L
        del tensors  # weights were repacked into the engine's own HBM buffers

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.wmar_gpt_destroy(h)
            self._h = None

    @property
    def device_bytes(self) -> int:
        return int(self._L.wmar_gpt_device_bytes(self._h))

    def decode_step(self, tok: torch.Tensor, pos: int) -> torch.Tensor:
        """One token per sequence at position `pos` -> logits [B, V]."""
        _require_cuda(tok, "tokens")
        tok = tok.to(torch.int64).contiguous().view(-1)
        B = tok.shape[0]
        logits = torch.empty(B, self.cfg.vocab_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_gpt_decode_step(self._h, tok.data_ptr(), B, int(pos), logits.data_ptr(),
                                                    _lib.stream_ptr(self.device)))
        return logits

    def generate(self, cond: torch.Tensor, steps: int, q: torch.Tensor, temperature=1.0, top_k=None, top_p=None,
                 wm_ctx: Optional[_lib.WmCtx] = None, use_graph: bool = True, trace_logits: bool = False):
        """sample_with_past: cond int64 [B], q float32 [steps, B, V] -> tokens int64 [B, steps]."""
        _require_cuda(cond, "conditioning")
        _require_cuda(q, "q")
        cond = cond.to(torch.int64).contiguous().view(-1)
        B = cond.shape[0]
        V = self.cfg.vocab_size
        assert q.shape == (steps, B, V) and q.dtype == torch.float32 and q.is_contiguous()
        out = torch.empty(B, steps, dtype=torch.int64, device=self.device)
        trace = torch.empty(steps, B, V, dtype=torch.float32, device=self.device) if trace_logits else None
        sp = _lib.SampleParams(float(temperature), int(top_k) if top_k else 0,
                               float(top_p) if top_p is not None else -1.0, 1 if use_graph else 0)
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_gpt_generate(
                self._h, C.byref(wm_ctx) if wm_ctx is not None else None, C.byref(sp), cond.data_ptr(), B, int(steps),
                q.data_ptr(), out.data_ptr(), trace.data_ptr() if trace is not None else None,
                _lib.stream_ptr(self.device)))
            # waits for the replays; raises if the fused projection launch's in-kernel barrier gave up (results invalid)
            _lib.check(self._L.wmar_gpt_check(self._h, _lib.stream_ptr(self.device)))
        return (out, trace) if trace_logits else out

    def profile_role(self, role: str, B: int, kv_len: int = 128, iters: int = 96) -> float:
        """Average microseconds per launch of one role's kernel replayed back to back (HIP events)."""
        out = C.c_double()
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_gpt_profile_role(self._h, self.T_CLASSES.index(role), B, kv_len, iters,
                                                     _lib.stream_ptr(self.device), C.byref(out)))
        return out.value

    def plan_info(self, B: int) -> Dict[str, str]:
        """{role: kernel} of a decode step at batch B, as the engine selects them."""
        buf = C.create_string_buffer(2048)
        _lib.check(self._L.wmar_gpt_plan_info(self._h, int(B), buf, len(buf)))
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";"))

    def set_attention_phases(self, one_wave_upto: int, two_waves_upto: int):
        """Cache lengths up to which the decode attention runs 1 / 2 waves per (sequence, head) (4 beyond)."""
        _lib.check(self._L.wmar_gpt_set_attention_phases(self._h, int(one_wave_upto), int(two_waves_upto)))

    def set_timing(self, on: bool):
        self._L.wmar_gpt_set_timing(self._h, 1 if on else 0)

    T_CLASSES = ["embed", "qkv", "attn", "proj", "resid", "fc1", "fc2", "head", "sample"]

    def get_timing(self):
        """{class: (total_us, calls)} of the last eager generate() with timing on, and ms/step."""
        n = len(self.T_CLASSES)
        us = (C.c_double * n)()
        calls = (C.c_int64 * n)()
        step = C.c_double()
        self._L.wmar_gpt_get_timing(self._h, us, calls, C.byref(step))
        return {k: (us[i], calls[i]) for i, k in enumerate(self.T_CLASSES)}, step.value


class VQGANEngine:
    """Taming VQGAN encode / decode; replaces VQModel.encode/decode + VectorQuantizer2
    (deps/taming/models/vqgan.py:64-73, modules/vqvae/quantize.py:272-331)."""

    def __init__(self, cfg: VQConfig, state: Dict[str, torch.Tensor], max_batch: int = 64, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        L = _lib.load()
        tensors = {k: v.detach().to(device=self.device, dtype=torch.float32).contiguous() for k, v in state.items()
                   if not k.startswith("loss.")}
        names, ptrs, n = _lib.tensor_table(tensors)
        c = _lib.VqConfig()
        c.ch, c.num_res_blocks, c.resolution = cfg.ch, cfg.num_res_blocks, cfg.resolution
        c.in_channels, c.out_ch, c.z_channels = cfg.in_channels, cfg.out_ch, cfg.z_channels
        c.embed_dim, c.n_embed, c.n_levels = cfg.embed_dim, cfg.n_embed, len(cfg.ch_mult)
        for i, m in enumerate(cfg.ch_mult):
            c.ch_mult[i] = m
        c.n_attn_res = len(cfg.attn_resolutions)
        for i, r in enumerate(cfg.attn_resolutions):
            c.attn_resolutions[i] = r
        c.max_batch = self.max_batch
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.wmar_vq_create(C.byref(c), names, ptrs, n, _lib.stream_ptr(self.device), C.byref(h)))
        self._h = h
        self._L = L

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.wmar_vq_destroy(h)
            self._h = None

    @property
    def device_bytes(self) -> int:
        return int(self._L.wmar_vq_device_bytes(self._h))

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        _require_cuda(codes, "codes")
        codes = codes.to(torch.int64).contiguous()
        B = codes.shape[0]
        R = self.cfg.resolution
        out = torch.empty(B, self.cfg.out_ch, R, R, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for b0 in range(0, B, self.max_batch):
                b1 = min(B, b0 + self.max_batch)
                _lib.check(self._L.wmar_vq_decode(self._h, codes[b0:b1].data_ptr(), b1 - b0, out[b0:b1].data_ptr(),
                                                  _lib.stream_ptr(self.device)))
        return out

    def encode(self, images: torch.Tensor, return_prequant: bool = False):
        _require_cuda(images, "images")
        images = images.to(torch.float32).contiguous()
        B = images.shape[0]
        S = self.cfg.codes_size
        codes = torch.empty(B, S * S, dtype=torch.int64, device=self.device)
        pre = torch.empty(B * S * S, self.cfg.embed_dim, dtype=torch.float32, device=self.device) if return_prequant else None
        with torch.cuda.device(self.device):
            for b0 in range(0, B, self.max_batch):
                b1 = min(B, b0 + self.max_batch)
                _lib.check(self._L.wmar_vq_encode(
                    self._h, images[b0:b1].data_ptr(), b1 - b0, codes[b0:b1].data_ptr(),
                    pre[b0 * S * S:b1 * S * S].data_ptr() if pre is not None else None, _lib.stream_ptr(self.device)))
        return (codes, pre) if return_prequant else codes


class RAREngine:
    """RAR generator with KV cache, adaLN, qk-norm and classifier-free guidance; replaces
    RAR.forward_fn / RAR.generate (deps/rar/modeling/rar.py:319-459)."""

    def __init__(self, cfg, state: Dict[str, torch.Tensor], max_batch: int = 64, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        L = _lib.load()
        tensors = {k: v.detach().to(device=self.device, dtype=torch.float32).contiguous() for k, v in state.items()
                   if k != "attn_mask"}
        names, ptrs, n = _lib.tensor_table(tensors)
        c = _lib.RarConfig(cfg.hidden_size, cfg.num_hidden_layers, cfg.num_attention_heads, cfg.intermediate_size,
                           cfg.image_seq_len, cfg.codebook_size, cfg.condition_num_classes, self.max_batch)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.wmar_rar_create(C.byref(c), names, ptrs, n, _lib.stream_ptr(self.device), C.byref(h)))
        self._h = h
        self._L = L

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.wmar_rar_destroy(h)
            self._h = None

    @property
    def device_bytes(self) -> int:
        return int(self._L.wmar_rar_device_bytes(self._h))

    def launch_status(self) -> Dict[str, int]:
        """{"fused": the fused residual + modulation launch is in use, "fallbacks": calls re-run on the two-launch pair}."""
        a, b = C.c_int32(), C.c_int32()
        _lib.check(self._L.wmar_rar_launch_status(self._h, C.byref(a), C.byref(b)))
        return {"fused": int(a.value), "fallbacks": int(b.value)}

    def forward_position(self, tok: torch.Tensor, cond_ids: torch.Tensor, pos: int) -> torch.Tensor:
        """tok int64 [M] (-1 = cls), cond_ids int64 [M] (offset condition ids) -> logits [M, V]."""
        _require_cuda(tok, "tokens")
        tok = tok.to(torch.int64).contiguous().view(-1)
        cond_ids = cond_ids.to(device=self.device, dtype=torch.int64).contiguous().view(-1)
        M = tok.shape[0]
        logits = torch.empty(M, self.cfg.codebook_size, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_rar_forward_position(self._h, tok.data_ptr(), cond_ids.data_ptr(), M, int(pos),
                                                         logits.data_ptr(), _lib.stream_ptr(self.device)))
        return logits

    def generate(self, class_ids: torch.Tensor, q: torch.Tensor, cfg_scales: Optional[torch.Tensor], temperature=1.0,
                 wm_ctx: Optional[_lib.WmCtx] = None, use_graph: bool = True) -> torch.Tensor:
        """class_ids int64 [B]; q float32 [L, B, V]; cfg_scales float32 [L] on the host (None: no guidance)."""
        _require_cuda(class_ids, "class ids")
        _require_cuda(q, "q")
        class_ids = class_ids.to(torch.int64).contiguous().view(-1)
        B = class_ids.shape[0]
        Ls, V = self.cfg.image_seq_len, self.cfg.codebook_size
        assert q.shape == (Ls, B, V) and q.dtype == torch.float32 and q.is_contiguous()
        out = torch.empty(B, Ls, dtype=torch.int64, device=self.device)
        sc = None
        if cfg_scales is not None:
            sc = cfg_scales.detach().to("cpu", torch.float32).contiguous()
            assert sc.numel() == Ls
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_rar_generate(
                self._h, C.byref(wm_ctx) if wm_ctx is not None else None, class_ids.data_ptr(), B,
                C.cast(sc.data_ptr(), C.POINTER(C.c_float)) if sc is not None else None, 1 if sc is not None else 0,
                float(temperature), q.data_ptr(), out.data_ptr(), 1 if use_graph else 0, _lib.stream_ptr(self.device)))
            _lib.check(self._L.wmar_rar_check(self._h, _lib.stream_ptr(self.device)))      # waits; raises if an in-launch wait gave up
        return out

    def generate_gumbel(self, class_ids, log_rs, cfg_scales, temperature=1.0, top_p=0.0, top_k=0, use_graph=True):
        """RAR.generate with the Gumbel-key sampler (extension, include/wmar_hip.h wmar_rar_generate_gumbel)."""
        _require_cuda(class_ids, "class ids")
        _require_cuda(log_rs, "gumbel key")
        class_ids = class_ids.to(torch.int64).contiguous().view(-1)
        B = class_ids.shape[0]
        Ls, V = self.cfg.image_seq_len, self.cfg.codebook_size
        assert log_rs.shape == (V,) and log_rs.dtype == torch.float32 and log_rs.is_contiguous()
        out = torch.empty(B, Ls, dtype=torch.int64, device=self.device)
        sc = None
        if cfg_scales is not None:
            sc = cfg_scales.detach().to("cpu", torch.float32).contiguous()
            assert sc.numel() == Ls
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_rar_generate_gumbel(
                self._h, class_ids.data_ptr(), B, C.cast(sc.data_ptr(), C.POINTER(C.c_float)) if sc is not None else None,
                1 if sc is not None else 0, float(temperature), float(top_p), int(top_k), log_rs.data_ptr(), out.data_ptr(),
                1 if use_graph else 0, _lib.stream_ptr(self.device)))
            _lib.check(self._L.wmar_rar_check(self._h, _lib.stream_ptr(self.device)))
        return out


class MaskgitVQEngine:
    """MaskGIT-VQGAN tokenizer of RAR; replaces PretrainedTokenizer.encode / decode_tokens
    (deps/rar/modeling/titok.py:75-89) incl. the wrapper's [-1,1] <-> [0,1] rescaling."""

    def __init__(self, cfg, state: Dict[str, torch.Tensor], max_batch: int = 64, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        L = _lib.load()
        tensors = {k: v.detach().to(device=self.device, dtype=torch.float32).contiguous() for k, v in state.items()
                   if k.startswith(("encoder.", "decoder.", "quantize."))}
        names, ptrs, n = _lib.tensor_table(tensors)
        c = _lib.MvqConfig()
        c.hidden_channels, c.num_res_blocks, c.resolution = cfg.hidden_channels, cfg.num_res_blocks, cfg.resolution
        c.num_channels, c.z_channels, c.num_embeddings = cfg.num_channels, cfg.z_channels, cfg.num_embeddings
        c.n_levels = len(cfg.channel_mult)
        for i, m in enumerate(cfg.channel_mult):
            c.channel_mult[i] = m
        c.max_batch = self.max_batch
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.wmar_mvq_create(C.byref(c), names, ptrs, n, _lib.stream_ptr(self.device), C.byref(h)))
        self._h = h
        self._L = L

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.wmar_mvq_destroy(h)
            self._h = None

    @property
    def device_bytes(self) -> int:
        return int(self._L.wmar_mvq_device_bytes(self._h))

    def decode(self, codes: torch.Tensor) -> torch.Tensor:
        _require_cuda(codes, "codes")
        codes = codes.to(torch.int64).contiguous()
        B = codes.shape[0]
        R = self.cfg.resolution
        out = torch.empty(B, self.cfg.num_channels, R, R, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            for b0 in range(0, B, self.max_batch):
                b1 = min(B, b0 + self.max_batch)
                _lib.check(self._L.wmar_mvq_decode(self._h, codes[b0:b1].data_ptr(), b1 - b0, out[b0:b1].data_ptr(),
                                                   _lib.stream_ptr(self.device)))
        return out

    def encode(self, images: torch.Tensor, return_prequant: bool = False):
        _require_cuda(images, "images")
        images = images.to(torch.float32).contiguous()
        B = images.shape[0]
        S = self.cfg.codes_size
        codes = torch.empty(B, S * S, dtype=torch.int64, device=self.device)
        pre = torch.empty(B * S * S, self.cfg.z_channels, dtype=torch.float32, device=self.device) if return_prequant else None
        with torch.cuda.device(self.device):
            for b0 in range(0, B, self.max_batch):
                b1 = min(B, b0 + self.max_batch)
                _lib.check(self._L.wmar_mvq_encode(
                    self._h, images[b0:b1].data_ptr(), b1 - b0, codes[b0:b1].data_ptr(),
                    pre[b0 * S * S:b1 * S * S].data_ptr() if pre is not None else None, _lib.stream_ptr(self.device)))
        return (codes, pre) if return_prequant else codes


class ChameleonEngine:
    """Chameleon / Anole transformer decode with KV cache, three guidance streams and the fused sampler; replaces
    ChameleonModelAdapter + Transformer.forward_with_attn_bias + the ImageDecoder token loop
    (deps/chameleon/inference/model_adapter.py:36-119, transformer.py:288-337, chameleon.py:299-389)."""

    def __init__(self, cfg, state: Dict[str, torch.Tensor], max_batch: int = 16, max_seq_len: int = 1024 + 128, device="cuda"):
        self.cfg = cfg
        self.device = torch.device(device)
        self.max_batch = int(max_batch)
        self.max_seq_len = int(max_seq_len)
        L = _lib.load()
        state = dict(state)
        for l in range(cfg.n_layers):   # the reference's load hooks (transformer.py:84-98, 197-208)
            p = f"layers.{l}."
            if p + "attention.wq.weight" in state:
                state[p + "attention.wqkv.weight"] = torch.cat([state.pop(p + "attention.wq.weight"), state.pop(p + "attention.wk.weight"),
                                                                 state.pop(p + "attention.wv.weight")])
            if p + "feed_forward.w1.weight" in state:
                state[p + "feed_forward.w13.weight"] = torch.cat([state.pop(p + "feed_forward.w1.weight"),
                                                                   state.pop(p + "feed_forward.w3.weight")])
        state.pop("rope.freqs", None)
        dts = {v.dtype for v in state.values()}
        bf16 = dts == {torch.bfloat16}
        dt = torch.bfloat16 if bf16 else torch.float32
        tensors = {k: v.detach().to(device=self.device, dtype=dt).contiguous() for k, v in state.items()}
        names, ptrs, n = _lib.tensor_table(tensors)
        c = _lib.ChamConfig(cfg.dim, cfg.n_layers, cfg.n_heads, cfg.n_kv_heads, cfg.vocab_size, cfg.ffn_hidden, cfg.norm_eps,
                            cfg.rope_theta, int(cfg.qk_normalization), int(cfg.swin_norm), 3 * self.max_batch, self.max_seq_len,
                            int(bf16))
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(L.wmar_cham_create(C.byref(c), names, ptrs, n, _lib.stream_ptr(self.device), C.byref(h)))
            torch.cuda.synchronize(self.device)
        self._h = h
        self._L = L

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._L.wmar_cham_destroy(h)
            self._h = None

    @property
    def device_bytes(self) -> int:
        return int(self._L.wmar_cham_device_bytes(self._h))

    def forward_tokens(self, tok: torch.Tensor, pos: torch.Tensor, want_logits: bool = True) -> Optional[torch.Tensor]:
        """tok int64 [M], pos int32 [M] -> logits float32 [M, V] (one token per sequence, warm caches)."""
        _require_cuda(tok, "tokens")
        tok = tok.to(torch.int64).contiguous().view(-1)
        pos = pos.to(device=self.device, dtype=torch.int32).contiguous().view(-1)
        M = tok.shape[0]
        logits = torch.empty(M, self.cfg.vocab_size, dtype=torch.float32, device=self.device) if want_logits else None
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_cham_forward_tokens(self._h, tok.data_ptr(), pos.data_ptr(), M,
                                                        logits.data_ptr() if want_logits else None, _lib.stream_ptr(self.device)))
        return logits

    def generate_image(self, prompts, q: torch.Tensor, n_tokens: int, temperature: float, top_p: Optional[float],
                       guidance_scale_text: float, guidance_scale_image: float, allow: Optional[torch.Tensor] = None,
                       wm_ctx: Optional[_lib.WmCtx] = None, use_graph: bool = True,
                       allow_ids: Optional[torch.Tensor] = None, pad_id: int = 1) -> torch.Tensor:
        """prompts: the 3B token lists (full-, image-, un-conditioned, in that order); q float32 [n_tokens, B, V];
        allow: int32 bitmap [V/32] of permitted vocabulary entries; allow_ids: the same set as ascending int32 ids (lets the
        sampler work on the compacted row); pad_id: the id short prompts are left-padded with (it can enter the watermark context
        of the first image tokens).  Returns int64 [B, n_tokens] vocabulary ids."""
        _require_cuda(q, "q")
        M = len(prompts)
        assert M % 3 == 0
        B = M // 3
        V = self.cfg.vocab_size
        assert q.shape == (n_tokens, B, V) and q.dtype == torch.float32 and q.is_contiguous()
        flat = np.ascontiguousarray(np.concatenate([np.asarray(p, dtype=np.int64) for p in prompts]))
        lens = np.ascontiguousarray(np.asarray([len(p) for p in prompts], dtype=np.int32))
        out = torch.empty(B, n_tokens, dtype=torch.int64, device=self.device)
        sp = _lib.ChamSampleParams(float(temperature), float(top_p) if top_p is not None else -1.0, float(guidance_scale_text),
                                   float(guidance_scale_image), 1 if use_graph else 0, int(pad_id))
        if allow is not None:
            _require_cuda(allow, "allow bitmap")
            assert allow.dtype == torch.int32 and allow.numel() == V // 32 and allow.is_contiguous()
        if allow_ids is not None:
            _require_cuda(allow_ids, "allow ids")
            assert allow is not None and allow_ids.dtype == torch.int32 and allow_ids.is_contiguous()
        with torch.cuda.device(self.device):
            _lib.check(self._L.wmar_cham_generate_image(
                self._h, C.byref(wm_ctx) if wm_ctx is not None else None, flat.ctypes.data, lens.ctypes.data, B, C.byref(sp),
                allow.data_ptr() if allow is not None else None, allow_ids.data_ptr() if allow_ids is not None else None,
                int(allow_ids.numel()) if allow_ids is not None else 0, q.data_ptr(), int(n_tokens), out.data_ptr(),
                _lib.stream_ptr(self.device)))
        return out
