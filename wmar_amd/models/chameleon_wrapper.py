"""Drop-in for ``wmar.models.chameleon_wrapper.ChameleonARMMWrapper`` on MI355X (wmar/models/chameleon_wrapper.py:16-186)
for the text -> image path (``sample`` / ``codes_to_images`` / ``images_to_codes``).

What the reference spreads over ``ChameleonInferenceModel`` + a worker thread + ``Generator`` / ``ImageDecoder`` /
``ChameleonGenerator`` / ``ChameleonModelAdapter`` (deps/chameleon/inference/chameleon.py:299-389, 392-440, 499-565,
generation.py:21-102, model_adapter.py:36-119) is one engine call here: prompts are tokenised on the host
(``TokenManager.tokens_from_ui``, chameleon.py:139-172), split into the three guidance streams (:351-372) and handed to
``ChameleonEngine.generate_image``; the worker thread, request queues and the watermarker-from-string round trip do not exist.
``sample_interleaved`` (text and image segments in one sequence) runs the same engine: eager text steps, captured image phases.
"""
from __future__ import annotations

import json
import os
from types import SimpleNamespace
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from ..utils.synth import CHAMELEON_7B, CHAMELEON_VQ, ChameleonConfig, VQConfig, synth_chameleon_state, synth_chameleon_vocab
from .armm_wrapper import AutoregressiveMultimodalModelWrapper
from .chameleon import VocabInfo, VocabTranslation
from .engine import ChameleonEngine, VQGANEngine
from .tokenizer_handles import ImageTokenizerHandle

_ASSETS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "assets")


def allow_bitmap(ids: Sequence[int], vocab_size: int, device) -> torch.Tensor:
    """int32 [V/32] bitmap of permitted vocabulary entries (AllowOnlyTokensLogitsProcessor, logits_processor.py:135-156)."""
    bits = np.zeros(vocab_size // 32, dtype=np.uint32)
    ids = np.asarray(list(ids), dtype=np.int64)
    np.bitwise_or.at(bits, ids >> 5, (np.uint32(1) << (ids & 31).astype(np.uint32)))
    return torch.from_numpy(bits.view(np.int32)).to(device)


def eight_bit_round_trip(images: torch.Tensor) -> torch.Tensor:
    """What the reference's ``images_to_codes`` does to the pixels before the VQGAN sees them (chameleon_wrapper.py:177-181):
    ``ImageTokenizer._pil_from_chw_tensor`` (image_tokenizer.py:100-122: clamp, (x+1)/2, *255, TRUNCATING uint8 cast) and
    ``_vqgan_input_from`` (:74-93: the 512x512 image passes resize / centre crop unchanged, then uint8/255 in float64, *2-1,
    float32).  Pure tensor ops on the tensor's own device -- no PIL image, no host copy."""
    x = (torch.clamp(images.to(torch.float32), -1.0, 1.0) + 1.0) / 2.0
    u8 = (x * 255).to(torch.uint8)
    return (u8.to(torch.float64) / 255.0 * 2 - 1).to(torch.float32)


class ChameleonARMMWrapper(AutoregressiveMultimodalModelWrapper):
    n_image_tokens = 1024

    def __init__(self, modelpath=None, seed=0, *, cfg: Optional[ChameleonConfig] = None, state: Optional[Dict[str, torch.Tensor]] = None,
                 vocab_map: Optional[Dict[str, int]] = None, tokenizer=None, vq_cfg: Optional[VQConfig] = None,
                 vq_state: Optional[Dict[str, torch.Tensor]] = None, device="cuda", max_batch=16, max_prompt_len=128):
        super().__init__()
        dev = torch.device(device)
        if modelpath is not None:
            from tokenizers import Tokenizer
            wdir = os.path.join(modelpath, "models", "7b")
            tpath = os.path.join(modelpath, "tokenizer", "text_tokenizer.json")
            params = json.load(open(os.path.join(wdir, "params.json")))
            params = {**params, **params.get("model", {}), **json.load(open(os.path.join(wdir, "consolidate_params.json")))}
            known = {f for f in ChameleonConfig.__dataclass_fields__}
            cfg = ChameleonConfig(**{k: v for k, v in params.items() if k in known})
            state = torch.load(os.path.join(wdir, "consolidated.pth"), map_location="cpu")
            tokenizer = Tokenizer.from_file(tpath)
            vocab_map = json.load(open(tpath))["model"]["vocab"]
            vq_cfg = CHAMELEON_VQ
            ck = torch.load(os.path.join(modelpath, "tokenizer", "vqgan_patched.ckpt"), map_location="cpu")
            vq_state = ck.get("state_dict", ck)
        assert cfg is not None and state is not None and vocab_map is not None and vq_cfg is not None and vq_state is not None
        self.vocab = VocabInfo(vocab_map)
        self.translation = VocabTranslation(self.vocab, device=dev)
        self.tokenizer = tokenizer
        self.seed = seed
        self.model = SimpleNamespace(device=dev, cfg=cfg, max_batch=max_batch, vocab=self.vocab,
                                     engine=ChameleonEngine(cfg, state, max_batch=max_batch,
                                                            max_seq_len=max_prompt_len + self.n_image_tokens, device=dev))
        self._vq_cfg = vq_cfg
        self._vq_state = {k: v.detach().to(dev, torch.float32) for k, v in vq_state.items() if not k.startswith("loss.")}
        self._vq_engine = None
        # token_manager.image_tokenizer._vq_model's place (chameleon_wrapper.py:44-45): handles on _vq_state
        self._image_tokenizer = ImageTokenizerHandle(self._vq_state, self._drop_vq_engine)
        ids = os.path.join(_ASSETS, "chameleon_all_ids.txt")
        if vq_cfg.n_embed == 8192 and cfg.vocab_size == 65536 and os.path.exists(ids):
            self.init_alivecodes(ids)
        else:   # synthetic vocabularies: every image token is alive, the rest of the vocabulary is dead
            vq = self.get_vq()
            alive = set(self.vocab.image_tokens)
            vq.alive_ids = torch.tensor(self.vocab.image_tokens, dtype=torch.long)
            vq.dead_ids = torch.tensor([t for t in range(cfg.vocab_size) if t not in alive], dtype=torch.long)
        self._allow_img = allow_bitmap(self.vocab.image_tokens, cfg.vocab_size, dev)
        self._allow_ids = torch.tensor(sorted(self.vocab.image_tokens), dtype=torch.int32, device=dev)
        # TextDecoder._allowed_tokens (chameleon.py:255-261) with txt and img both on
        self._allow_text = torch.tensor([self.vocab.eos_id] + self.vocab.text_tokens + [self.vocab.begin_image], dtype=torch.int64, device=dev)
        self.codes_size = vq_cfg.codes_size
        self.image_size = vq_cfg.resolution
        self.dim_z = vq_cfg.z_channels
        self.n_image_tokens = self.codes_size * self.codes_size
        self.watermarker = None
        self.watermarker_text = None
        self.use_graph = True
        self.guidance_scale_text, self.guidance_scale_image = 3.0, 1.2      # Options.Image.CFG defaults (chameleon.py:69-72)

    @classmethod
    def synthetic(cls, cfg=CHAMELEON_7B, vq_cfg=CHAMELEON_VQ, seed=0, device="cuda", max_batch=16, logit_scale=8.0, n_img=None):
        from ..utils.synth import synth_vq_state_fast
        sd = synth_chameleon_state(cfg, seed, device, logit_scale, gen_device=device)
        vm = synth_chameleon_vocab(cfg.vocab_size, n_img or vq_cfg.n_embed)
        return cls(None, seed, cfg=cfg, state=sd, vocab_map=vm, vq_cfg=vq_cfg, vq_state=synth_vq_state_fast(vq_cfg, seed, device),
                   device=device, max_batch=max_batch)

    def __repr__(self):
        return "ChameleonARMMWrapper"

    def _drop_vq_engine(self):
        self._vq_engine = None  # repacked from _vq_state on next use

    @property
    def vq_engine(self) -> VQGANEngine:
        if self._vq_engine is None:
            self._vq_engine = VQGANEngine(self._vq_cfg, self._vq_state, max_batch=min(self.model.max_batch, 16), device=self.model.device)
        return self._vq_engine

    def set_watermarker(self, watermarker=None, watermarker_text=None):
        if watermarker is not None and getattr(getattr(watermarker, "seed_strategy", None), "value", None) == "spatial":
            raise ValueError("Chameleon supports fixed / linear seeding only (the reference's generate.py asserts the same)")
        self.watermarker = watermarker
        self.watermarker_text = watermarker_text

    def get_image_tokenizer(self):
        return self._image_tokenizer

    def get_vq(self):
        return self.get_image_tokenizer().quantize

    def get_total_vocab_size(self):
        return len(self.vocab.all_tokens)

    # ---- prompt handling
    def tokens_from_ui(self, inputs: List[dict]) -> List[int]:
        """TokenManager.tokens_from_ui (chameleon.py:139-172) for text / sentinel / ids entries."""
        tokens = [self.vocab.bos_id]
        for inp in inputs:
            if inp["type"] == "text":
                if self.tokenizer is None:
                    raise RuntimeError("no text tokenizer loaded: pass prompts as {'type': 'ids', 'value': [...]}")
                tokens += self.tokenizer.encode(inp["value"]).ids
            elif inp["type"] == "sentinel":
                tokens += [{"<START-OF-IMAGE>": self.vocab.begin_image, "<END-OF-TURN>": self.vocab.eot_id}[inp["value"]]]
            elif inp["type"] == "ids":
                tokens += list(inp["value"])
            else:
                raise ValueError("Unknown input type.")
        return tokens

    def split_inputs_for_cfg(self, input_ids: List[List[int]]) -> List[List[int]]:
        """ImageDecoder.__init__ + _split_inputs_for_cfg (chameleon.py:329-372): append <racm3:break>, then the
        full-conditioned, image-conditioned (image tokens, bos, boi, eoi only) and unconditioned ([bos, boi]) streams."""
        v = self.vocab
        full = [list(s) + ([] if s and s[-1] == v.begin_image else [v.begin_image]) for s in input_ids]
        keep = set(v.image_tokens) | {v.bos_id, v.begin_image, v.end_image}
        img = [[t for t in s if t in keep] for s in full]
        unc = [[v.bos_id, v.begin_image] for _ in full]
        return full + img + unc


    # ------------------------------------------------------------------ interleaved text + image mode
    @staticmethod
    def split_token_sequence(tokens: torch.LongTensor, boi: int, eoi: int):
        """wmar/models/chameleon_wrapper.py:47-104: cut ONE generated sequence [1, n] into ("text_seg" | "image_seg", tokens [1, m])
        pieces at the begin-of-image / end-of-image markers (the markers themselves are dropped)."""
        batch_size, _ = tokens.shape
        assert batch_size == 1, "Batch size must be 1"
        device, dtype = tokens.device, tokens.dtype
        segments, current, in_image = [], [], False

        def flush(kind):
            segments.append((kind, torch.tensor(current, dtype=dtype, device=device).reshape(1, -1)))

        for token in tokens[0].tolist():
            if token == boi:
                if current:
                    flush("text_seg")
                    current = []
                in_image = True
            elif token == eoi and in_image:
                flush("image_seg")
                current = []
                in_image = False
            else:
                current.append(token)
        if current:
            flush("image_seg" if in_image else "text_seg")
        return segments

    def text_logits_chain(self, input_ids: torch.Tensor, logits: torch.Tensor, q: torch.Tensor, *, temperature: float = 0.7,
                          top_p: float = 0.9, repetition_penalty: float = 1.2, boi_limit: Optional[int] = None,
                          apply_watermark: bool = False, allowed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One text-mode step of TextDecoder (chameleon.py:262-284 + generation.py:84-93) on the device: text watermark (called
        positionally on the padded input rows) -> allow-only (eos + text tokens + <boi>) -> <boi> forbidden at or after
        `boi_limit` (max_seq_len - 1026) -> HF RepetitionPenaltyLogitsProcessor over the input rows -> /temperature -> top-p ->
        softmax -> multinomial as argmax(p / q).  input_ids int64 [B, t]; logits float32 [B, V] (modified in place); q [B, V]."""
        import ctypes as C
        from .. import _lib
        dev = self.model.device
        B, V = logits.shape
        if apply_watermark and self.watermarker_text is not None:
            logits = self.watermarker_text.spawn_logit_processor()(input_ids, logits)
        if allowed is None:
            allowed = self._allow_text
        mask = torch.ones(V, dtype=torch.bool, device=dev)
        mask[allowed] = False
        logits.masked_fill_(mask[None, :], float("-inf"))
        if boi_limit is not None and input_ids.shape[1] >= boi_limit:
            logits[:, self.vocab.begin_image] = float("-inf")
        if repetition_penalty != 1.0:
            score = torch.gather(logits, 1, input_ids)
            score = torch.where(score < 0, score * repetition_penalty, score / repetition_penalty)
            logits.scatter_(1, input_ids, score)
        tok = torch.empty(B, dtype=torch.int64, device=dev)
        scratch = torch.empty_like(logits)
        with torch.cuda.device(dev):
            _lib.check(_lib.load().wmar_sample_fused(None, logits.data_ptr(), B, V, None, 0, 0, float(temperature), 0,
                                                     float(top_p) if top_p is not None else -1.0, q.contiguous().data_ptr(),
                                                     scratch.data_ptr(), tok.data_ptr(), _lib.stream_ptr(dev)))
        return tok

    def _prefill(self, rows: List[List[int]]) -> torch.Tensor:
        """Right-aligned prefill (alignment.py:27-41 as the model adapter consumes it): logits of the last prompt position."""
        maxlen = max(len(r) for r in rows)
        lg = None
        for j in range(maxlen):
            tok = [r[j - (maxlen - len(r))] if j - (maxlen - len(r)) >= 0 else 0 for r in rows]
            pos = [max(j - (maxlen - len(r)), 0) for r in rows]
            lg = self.model.engine.forward_tokens(torch.tensor(tok, device=self.model.device),
                                                  torch.tensor(pos, dtype=torch.int32, device=self.model.device), want_logits=j == maxlen - 1)
        return lg

    # conditioning: list of (index, prompt) tuples; returns the segments of the ONE generated sequence (reference: batch 1)
    def sample_interleaved(self, conditioning, gen_params, apply_watermark=False, max_gen_len: int = 4096,
                           text_temperature: float = 0.7, text_top_p: float = 0.9, repetition_penalty: float = 1.2):
        """wmar/models/chameleon_wrapper.py:108-134 -> Generator with Options(txt=True) (chameleon.py:392-440): text tokens are
        decoded until every row emits <boi>, then 1024 image tokens under 3-way guidance, <eoi>, text again ... until <eos> or the
        length limit.  Every switch re-runs the prompt through a fresh decoder, exactly as the reference builds a new
        TextDecoder / ImageDecoder on the grown input.  The image phases are the captured engine loop; the text steps are eager
        (one engine forward + the text chain per token)."""
        v = self.vocab
        eng = self.model.engine
        # the reference's interleaved mode serves one prompt per call (its split_token_sequence asserts batch 1): fail at entry, not
        # after the whole generation
        assert len(conditioning) == 1, f"sample_interleaved takes one prompt per call (got {len(conditioning)})"
        rows = []
        for _, prompt in conditioning:
            item = {"type": "text", "value": prompt} if isinstance(prompt, str) else {"type": "ids", "value": prompt}
            rows.append(self.tokens_from_ui([item, {"type": "sentinel", "value": "<END-OF-TURN>"}]))
        B = len(rows)
        if self.seed is not None:
            torch.manual_seed(self.seed)
        max_seq_len = eng.max_seq_len
        generated: List[List[int]] = [[] for _ in range(B)]
        V = self.model.cfg.vocab_size
        wm_ctx = self.watermarker.wm_ctx() if (apply_watermark and self.watermarker is not None) else None
        done = False
        while not done:
            # ---------------- text decoder on the current inputs
            inputs = [r + g for r, g in zip(rows, generated)]
            max_prompt_len = max(len(r) for r in inputs)
            limit = min(max_seq_len, max_prompt_len + max_gen_len)
            padded = torch.tensor([[v.pad_id] * (max_prompt_len - len(r)) + r for r in inputs], dtype=torch.int64, device=self.model.device)
            lens = torch.tensor([len(r) for r in inputs], dtype=torch.int32, device=self.model.device)
            lg = self._prefill(inputs)
            switch = False
            n_new = 0
            while True:
                # stopping criteria of the TextDecoder (chameleon.py:232-236): length, or <eos> in every row after the prompt
                if padded.shape[1] >= limit or bool((padded[:, max_prompt_len:] == v.eos_id).any(dim=1).all()):
                    done = True
                    break
                q = self._noise_draw(lambda t, g: t.exponential_(1, generator=g), (B, V))
                tok = self.text_logits_chain(padded, lg, q, temperature=text_temperature, top_p=text_top_p,
                                             repetition_penalty=repetition_penalty, boi_limit=max_seq_len - (self.n_image_tokens + 2),   # max_seq_len - 1026 at 1024 image tokens (chameleon.py:271-275)
                                             apply_watermark=apply_watermark)
                padded = torch.cat([padded, tok[:, None]], dim=1)
                for b, t in enumerate(tok.tolist()):
                    generated[b].append(t)
                n_new += 1
                if bool((tok == v.begin_image).all()):
                    switch = True
                    break
                lg = eng.forward_tokens(tok, lens + (n_new - 1))
            if done or not switch:
                break
            # ---------------- image decoder: 1024 tokens, then <eoi> (chameleon.py:374-389)
            inputs = [r + g for r, g in zip(rows, generated)]
            if max(len(r) for r in inputs) + self.n_image_tokens > max_seq_len:
                break
            q = self.draw_noise(B)
            img = eng.generate_image(self.split_inputs_for_cfg(inputs), q, self.n_image_tokens, gen_params["temperature"], gen_params["top_p"],
                                     self.guidance_scale_text, self.guidance_scale_image, allow=self._allow_img, wm_ctx=wm_ctx,
                                     use_graph=self.use_graph, allow_ids=self._allow_ids, pad_id=v.pad_id)
            for b in range(B):
                generated[b] += img[b].tolist() + [v.end_image]
        codes = torch.tensor(generated, dtype=torch.int64, device=self.model.device).contiguous()
        return self.split_token_sequence(codes, v.begin_image, v.end_image)

    def draw_noise(self, B: int, generator=None) -> torch.Tensor:
        """One [B, V] Exp(1) draw per image token: what ``probs.multinomial`` on the first stream consumes (token_selector.py:26-47)."""
        V = self.model.cfg.vocab_size
        q = torch.empty(self.n_image_tokens, B, V, dtype=torch.float32, device=self.model.device)
        for n in range(self.n_image_tokens):
            q[n].copy_(self._noise_draw(lambda t, g: t.exponential_(1, generator=g), (B, V), generator))
        return q

    # conditioning: list of (index, prompt) tuples (prompt: str, or a list of token ids); gen_params: {top_p, temperature}
    def sample(self, conditioning, gen_params, apply_watermark=False, q: Optional[torch.Tensor] = None):
        prompts = []
        for _, prompt in conditioning:
            item = {"type": "text", "value": prompt} if isinstance(prompt, str) else {"type": "ids", "value": prompt}
            prompts.append(self.tokens_from_ui([item, {"type": "sentinel", "value": "<END-OF-TURN>"}]))
        B = len(prompts)
        dev = self.model.device
        out = torch.empty(B, self.n_image_tokens, dtype=torch.int64, device=dev)
        wm_ctx = self.watermarker.wm_ctx() if (apply_watermark and self.watermarker is not None) else None
        if q is None and self.seed is not None:
            torch.manual_seed(self.seed)          # enable_full_determinism(options.seed) at Generator start (chameleon.py:402-403)
        mb = self.model.max_batch
        for b0 in range(0, B, mb):
            b1 = min(B, b0 + mb)
            qq = q[:, b0:b1].contiguous() if q is not None else self.draw_noise(b1 - b0)
            out[b0:b1] = self.model.engine.generate_image(
                self.split_inputs_for_cfg(prompts[b0:b1]), qq, self.n_image_tokens, gen_params["temperature"], gen_params["top_p"],
                self.guidance_scale_text, self.guidance_scale_image, allow=self._allow_img, wm_ctx=wm_ctx, use_graph=self.use_graph,
                allow_ids=self._allow_ids, pad_id=self.vocab.pad_id)
        codes = out.detach().contiguous()
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        return codes

    # codes: [b, 1024] BPE ids of image tokens -> [b, 3, 512, 512] pixels in [-1, 1]
    def codes_to_images(self, codes):
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        img_ids = self.translation.convert_bpe2img(codes.to(self.model.device))
        images = self.vq_engine.decode(img_ids).clamp(-1, 1)
        assert self.is_images_shaped(images), f"Images shape: {images.shape}"
        return images

    # images -> BPE ids; the reference goes through an 8-bit PIL image (truncating cast, image_tokenizer.py:100-122, :74-86)
    def images_to_codes(self, images):
        assert self.is_images_shaped(images), f"Images shape: {images.shape}"
        x = eight_bit_round_trip(images.to(self.model.device))
        codes = self.translation.convert_img2bp2(self.vq_engine.encode(x)).to(torch.int64)
        assert self.is_codes_shaped(codes), f"Codes shape: {codes.shape}"
        return codes
