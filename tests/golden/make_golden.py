"""Generate tests/golden/*.npz|json by RUNNING THE REFERENCE in this container.

Usage (build container only; the GPU box never runs this):
    python tests/golden/make_golden.py [--ref /root/reference]

The reference is imported from its checkout (never copied).  Packages it imports
that are absent here (loguru, pytorch_lightning, torchvision) are replaced by
empty stand-in modules fabricated in a temp dir -- they carry no reference code.
Everything written is data: inputs and the reference's outputs.
"""
import argparse
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)


def _stubs(d):
    os.makedirs(os.path.join(d, "loguru"))
    open(os.path.join(d, "loguru", "__init__.py"), "w").write(
        "class _L:\n    def __getattr__(s, n):\n        return lambda *a, **k: None\nlogger = _L()\n")
    os.makedirs(os.path.join(d, "pytorch_lightning"))
    open(os.path.join(d, "pytorch_lightning", "__init__.py"), "w").write(
        "import torch.nn as nn\nclass LightningModule(nn.Module):\n    @property\n    def device(self):\n"
        "        return next(self.parameters()).device\n")
    for sub in ("", "transforms", "models"):
        os.makedirs(os.path.join(d, "torchvision", sub), exist_ok=True)
    open(os.path.join(d, "torchvision", "__init__.py"), "w").write("")
    open(os.path.join(d, "torchvision", "models", "__init__.py"), "w").write("")
    open(os.path.join(d, "torchvision", "transforms", "__init__.py"), "w").write(
        "class PILToTensor:\n    pass\nfrom . import functional\n")
    open(os.path.join(d, "torchvision", "transforms", "functional.py"), "w").write("")
    # timm.layers.Mlp (third-party, absent here): its published semantics fc1 -> act -> drop -> norm -> fc2 -> drop
    os.makedirs(os.path.join(d, "timm", "layers"))
    open(os.path.join(d, "timm", "__init__.py"), "w").write("")
    open(os.path.join(d, "timm", "layers", "__init__.py"), "w").write(
        "import torch.nn as nn\n"
        "class Mlp(nn.Module):\n"
        "    def __init__(self, in_features, hidden_features=None, out_features=None, act_layer=nn.GELU, norm_layer=None,\n"
        "                 bias=True, drop=0.0, use_conv=False):\n"
        "        super().__init__()\n"
        "        out_features = out_features or in_features\n"
        "        hidden_features = hidden_features or in_features\n"
        "        self.fc1 = nn.Linear(in_features, hidden_features, bias=bias)\n"
        "        self.act = act_layer()\n"
        "        self.drop1 = nn.Dropout(drop)\n"
        "        self.norm = norm_layer(hidden_features) if norm_layer is not None else nn.Identity()\n"
        "        self.fc2 = nn.Linear(hidden_features, out_features, bias=bias)\n"
        "        self.drop2 = nn.Dropout(drop)\n"
        "    def forward(self, x):\n"
        "        return self.drop2(self.fc2(self.norm(self.drop1(self.act(self.fc1(x))))))\n")


def rar_vectors(args, synth, wms, rs):
    """RAR generator + MaskGIT-VQGAN tokenizer vectors."""
    import torch
    from deps.rar.modeling.modules.maskgit_vqgan import Decoder, Encoder, VectorQuantizer
    from deps.rar.modeling.rar import RAR

    class AD(dict):
        def __getattr__(self, k):
            v = self[k]
            return AD(v) if isinstance(v, dict) else v

        def get(self, k, d=None):
            return dict.get(self, k, d)

    out = {}
    rcfg = synth.RARConfig(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                           image_seq_len=16, codebook_size=1024, condition_num_classes=1000)
    cfg = AD(model=dict(vq_model=dict(codebook_size=1024),
                        generator=dict(hidden_size=128, num_hidden_layers=2, num_attention_heads=4, intermediate_size=512,
                                       image_seq_len=16, condition_num_classes=1000, dropout=0.0, attn_drop=0.0)))
    sd = synth.synth_rar_state(rcfg, seed=2, logit_scale=30.0)
    gen = RAR(cfg).eval()
    gen.load_state_dict(sd, strict=True)   # pins the checkpoint key layout (attn_mask is non-persistent)
    gen.set_random_ratio(0)
    wm = wms["rar"]
    wm.delta = 2.0
    cond = torch.tensor([[3], [977], [0], [512]], dtype=torch.long)
    rec = []
    orig = gen.forward_fn

    def spy(ids, condition, **kw):
        r = orig(ids, condition, **kw)
        rec.append(r[:, -1].detach().clone().numpy())
        return r

    gen.forward_fn = spy
    torch.manual_seed(21)
    toks = gen.generate(condition=cond, guidance_scale=4.0, guidance_scale_pow=0.0, randomize_temperature=1.0,
                        logit_processor=wm.spawn_logit_processor())
    out["rar_cond"] = cond.view(-1).numpy()
    out["rar_tokens_wm"] = toks.numpy()
    out["rar_logits"] = np.stack(rec)[:5]           # [5, 2B, V] cond rows then uncond rows, steps 0..4
    rec.clear()
    torch.manual_seed(21)
    out["rar_tokens_nowm"] = gen.generate(condition=cond, guidance_scale=4.0, guidance_scale_pow=0.0,
                                          randomize_temperature=1.0).numpy()
    rec.clear()
    torch.manual_seed(21)
    out["rar_tokens_pow"] = gen.generate(condition=cond, guidance_scale=3.0, guidance_scale_pow=1.5,
                                         randomize_temperature=0.9, logit_processor=wm.spawn_logit_processor()).numpy()
    gen.forward_fn = orig
    torch.manual_seed(21)
    torch.rand(4, 1)   # the label-drop mask draw of preprocess_condition (rar.py:305) precedes the sampling noise
    out["rar_q"] = np.stack([torch.empty(4, 1024).exponential_(1).numpy() for _ in range(16)])[:5]
    out["rar_pvals_wm"] = wm.detect(toks).numpy()

    # tokenizer
    mcfg = synth.MaskgitVQConfig(hidden_channels=32, channel_mult=(1, 2, 2), num_res_blocks=1, resolution=32, z_channels=16,
                                 num_embeddings=256)
    msd = synth.synth_maskgit_state(mcfg, seed=4)
    tc = AD(channel_mult=list(mcfg.channel_mult), num_resolutions=mcfg.num_resolutions, dropout=0.0,
            hidden_channels=mcfg.hidden_channels, num_channels=3, num_res_blocks=mcfg.num_res_blocks,
            resolution=mcfg.resolution, z_channels=mcfg.z_channels)
    enc, dec = Encoder(tc).eval(), Decoder(tc).eval()
    vq = VectorQuantizer(mcfg.num_embeddings, mcfg.z_channels, 0.25).eval()
    enc.load_state_dict({k[8:]: v for k, v in msd.items() if k.startswith("encoder.")}, strict=True)
    dec.load_state_dict({k[8:]: v for k, v in msd.items() if k.startswith("decoder.")}, strict=True)
    vq.load_state_dict({"embedding.weight": msd["quantize.embedding.weight"]}, strict=True)
    S = mcfg.codes_size
    codes = torch.from_numpy(rs.randint(0, mcfg.num_embeddings, size=(2, S * S)).astype(np.int64))
    with torch.no_grad():
        img01 = torch.clamp(dec(vq.get_codebook_entry(codes)), 0.0, 1.0)          # titok.py:81-85
        img = torch.clamp(img01 * 2.0 - 1.0, -1.0, 1.0)                            # rar_wrapper.py:112-114
        h = enc((img + 1.0) / 2.0)                                                  # rar_wrapper.py:124-125
        _, idx, _ = vq(h)
    out["mg_codes"] = codes.numpy()
    out["mg_images"] = img.numpy()
    out["mg_prequant"] = h.permute(0, 2, 3, 1).reshape(-1, mcfg.z_channels).numpy()
    out["mg_codes_roundtrip"] = idx.view(2, -1).numpy()
    np.savez_compressed(os.path.join(HERE, "rar_vectors.npz"), **out)
    print("wrote rar_vectors.npz", os.path.getsize(os.path.join(HERE, "rar_vectors.npz")), "bytes")


def synth_vocab_map(n_vocab=2048, n_img=512, first_img=4):
    """A Chameleon-style vocabulary (tokenizer json: name -> id): specials, IMGIMG<letters>Z image tokens (vocab.py:77-93), text."""
    vm = {"<pad>": 1, "<s>": 0, "</s>": 2, "<reserved08706>": 3}
    digits = "ABCDEFGHIJ"
    nxt = first_img
    for code in range(n_img):
        vm["IMGIMG" + "".join(digits[int(c)] for c in str(code)) + "Z"] = nxt
        nxt += 1
    vm["<racm3:break>"] = nxt
    vm["<eoss>"] = nxt + 1
    nxt += 2
    i = 0
    while nxt < n_vocab:
        vm[f"t{i}"] = nxt
        nxt += 1
        i += 1
    return vm


def chameleon_vectors(args):
    """Known answers of the importable Chameleon host pieces: logits processors (logits_processor.py:135-156, 312-336),
    HF temperature / top-p warpers as chameleon.py:313-327 chains them, MultinomialTokenSelector / ReplicatedInputTokenSelector
    (token_selector.py:26-47), VocabInfo / VocabTranslation (vocab.py), the GentimeWatermark positional call."""
    import torch
    from transformers import LogitsProcessorList, TemperatureLogitsWarper, TopPLogitsWarper
    from deps.chameleon.inference.logits_processor import AllowOnlyTokensLogitsProcessor, InBatchInstructCFGLogitsProcessor
    from deps.chameleon.inference.token_selector import MultinomialTokenSelector, ReplicatedInputTokenSelector
    from deps.chameleon.inference.vocab import VocabInfo, VocabTranslation
    from wmar.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy

    out = {}
    vm = synth_vocab_map()
    V = len(vm)
    vi = VocabInfo(vm)
    vt = VocabTranslation(vi)
    out["cham_image_tokens"] = np.array(vi.image_tokens, dtype=np.int64)
    out["cham_text_tokens"] = np.array(vi.text_tokens, dtype=np.int64)
    out["cham_special_tokens"] = np.array(vi.special_tokens, dtype=np.int64)
    out["cham_ids"] = np.array([vi.bos_id, vi.eos_id, vi.boi_id, vi.eoi_id, vi.pad_id, vi.eot_id], dtype=np.int64)
    bpe = torch.tensor([[4, 5, 515, 100], [300, 4, 4, 17]])
    out["cham_bpe_batch"] = bpe.numpy()
    out["cham_bpe2img"] = vt.convert_bpe2img(bpe).numpy()
    img = torch.tensor([[0, 511, 9], [10, 99, 100]])
    out["cham_img_batch"] = img.numpy()
    out["cham_img2bpe"] = vt.convert_img2bp2(img).numpy().astype(np.int64)

    B = 5
    g = torch.Generator().manual_seed(77)
    logits = (torch.randn(3 * B, V, generator=g) * 2.5).bfloat16().float()      # bf16-valued like the model's output
    out["cham_logits3"] = logits.numpy()
    alive = vi.image_tokens
    dead = sorted(set(range(V)) - set(alive))
    vq = {"alive_ids": torch.tensor(alive), "dead_ids": torch.tensor(dead), "embedding": torch.zeros(V, 4)}
    input_ids = torch.randint(0, V, (3 * B, 7), generator=g)
    input_ids[:, -1] = vi.boi_id
    out["cham_input_ids"] = input_ids.numpy()
    for name, seed_s, h, delta, temp, top_p in [("fixed", SeedStrategy.FIXED, 0, 2.0, 1.0, 0.9),
                                                ("linear", SeedStrategy.LINEAR, 1, 4.0, 0.7, 0.5),
                                                ("nowm", None, 0, 0.0, 1.0, 0.9)]:
        procs = [InBatchInstructCFGLogitsProcessor(3.0, 1.2)]
        if seed_s is not None:
            wm = GentimeWatermark(vq, V, seed_s, SplitStrategy.RANDOM_STRATIFIED, h, delta, 0.25, device="cpu")
            procs.append(wm.spawn_logit_processor())
        procs += [AllowOnlyTokensLogitsProcessor(vi.image_tokens), TemperatureLogitsWarper(temp), TopPLogitsWarper(top_p)]
        lp = LogitsProcessorList(procs)
        lg = lp(input_ids, logits.clone())
        probs = lg.softmax(dim=1)
        sel = ReplicatedInputTokenSelector(MultinomialTokenSelector(), n=3)
        torch.manual_seed(1234)
        tok = sel(input_ids, probs)
        torch.manual_seed(1234)
        q = torch.empty(B, V).exponential_(1)
        assert torch.equal(tok[:B], (probs[:B] / q).argmax(dim=1)) and torch.equal(tok, tok[:B].repeat(3))
        out[f"cham_{name}_processed"] = lg[:B].numpy()
        out[f"cham_{name}_tok"] = tok[:B].numpy().astype(np.int64)
        out[f"cham_{name}_q"] = q.numpy()
        out[f"cham_{name}_params"] = np.array([h, delta, temp, top_p], dtype=np.float64)
    # ---- text-mode logits chain of TextDecoder._logits_processors (chameleon.py:262-284) for the interleaved mode: text
    # watermark -> allow-only (eos + text tokens + <boi>) -> DisallowTokensAtOrAfterIndex([boi], max_seq_len - 1026) ->
    # HF RepetitionPenalty(1.2) -> Temperature(0.7) -> TopP(0.9) -> MultinomialTokenSelector (token_selector.py:26-31)
    from transformers import RepetitionPenaltyLogitsProcessor
    from deps.chameleon.inference.logits_processor import DisallowTokensAtOrAfterIndexLogitsProcessor
    Bt = 6
    gt = torch.Generator().manual_seed(78)                         # its own stream: the fixtures below keep their values
    tl = (torch.randn(Bt, V, generator=gt) * 2.5).bfloat16().float()
    tin = torch.randint(0, V, (Bt, 9), generator=gt)
    tin[0, :3] = vi.pad_id                                         # left padding of a shorter prompt
    tin[1, 4] = tin[1, 7]                                           # a repeated token
    allowed = [vi.eos_id] + vi.text_tokens + [vi.begin_image]
    out["cham_text_logits"] = tl.numpy()
    out["cham_text_input_ids"] = tin.numpy()
    out["cham_text_allowed"] = np.array(allowed, dtype=np.int64)
    for name, seed_s, h, limit in [("wm", SeedStrategy.LINEAR, 1, 100), ("nowm_limit", None, 0, 9), ("fixed", SeedStrategy.FIXED, 0, 100)]:
        procs = []
        if seed_s is not None:
            wmt = GentimeWatermark(vq, V, seed_s, SplitStrategy.RANDOM_STRATIFIED, h, 2.0, 0.25, device="cpu")
            procs.append(wmt.spawn_logit_processor())
        procs += [AllowOnlyTokensLogitsProcessor(allowed), DisallowTokensAtOrAfterIndexLogitsProcessor([vi.begin_image], limit),
                  RepetitionPenaltyLogitsProcessor(1.2), TemperatureLogitsWarper(0.7), TopPLogitsWarper(0.9)]
        lgt = LogitsProcessorList(procs)(tin, tl.clone())
        probs = lgt.softmax(dim=1)
        torch.manual_seed(4321)
        tokt = MultinomialTokenSelector()(tin, probs)
        torch.manual_seed(4321)
        qt = torch.empty(Bt, V).exponential_(1)
        assert torch.equal(tokt, (probs / qt).argmax(dim=1))
        out[f"cham_text_{name}_processed"] = lgt.numpy()
        out[f"cham_text_{name}_tok"] = tokt.numpy().astype(np.int64)
        out[f"cham_text_{name}_limit"] = np.array(limit)
    out["cham_text_q"] = qt.numpy()
    # ---- the image tokenizer side: Chameleon's own VQGAN (deps/chameleon/inference/vqgan.py) on a reduced config with the
    # released model's topology (no attention in the down/up path, one in the middle).  image_tokenizer.py itself does not
    # import under this Python (it annotates with the PIL.Image MODULE inside typing.Union).
    from deps.chameleon.inference.vqgan import VQModel
    from wmar_amd.utils import synth
    vcfg = synth.VQConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(), resolution=32, z_channels=32, embed_dim=32,
                          n_embed=256)
    vsd = synth.synth_vq_state(vcfg, seed=5)
    vq = VQModel(ddconfig=dict(double_z=False, z_channels=32, resolution=32, in_channels=3, out_ch=3, ch=32, ch_mult=[1, 2, 2],
                               num_res_blocks=1, attn_resolutions=[], dropout=0.0), n_embed=256, embed_dim=32).eval()
    vq.load_state_dict(vsd, strict=True)
    codes = torch.randint(0, 256, (2, 64), generator=g)
    with torch.no_grad():
        zq = vq.quantize.get_codebook_entry(codes.view(-1), (2, 8, 8, 32))
        imgs = vq.decode(zq)
        h = vq.quant_conv(vq.encoder(imgs))
        _, _, (_, _, idx) = vq.encode(imgs)
    out["chvq_codes"] = codes.numpy()
    out["chvq_images"] = imgs.numpy()
    out["chvq_prequant"] = h.permute(0, 2, 3, 1).reshape(-1, 32).numpy()
    out["chvq_codes_roundtrip"] = idx.view(2, 64).numpy().astype(np.int64)
    np.savez_compressed(os.path.join(HERE, "chameleon_vectors.npz"), **out)
    print("wrote chameleon_vectors.npz", os.path.getsize(os.path.join(HERE, "chameleon_vectors.npz")), "bytes")


def gumbel_vectors(args):
    """Known answers of wmar_audio/watermark/engine.py (gumbel_sample, gumbel_score_tok, get_wm_window_hash)."""
    import torch
    sys.path.insert(0, os.path.join(args.ref, "wmar_audio"))
    import moshi  # noqa: F401  (engine.py imports moshi.utils.sampling)
    from watermark import engine as E

    out = {}
    V, B = 1024, 12
    g = torch.Generator().manual_seed(99)
    logits = (torch.randn(B, V, generator=g) * 3.0).float()
    logits[3] = logits[3] * 4.0            # a peaked row
    logits[5, 10] = logits[5, 700]         # an exact tie between two logits
    out["gum_logits"] = logits.numpy()
    same = E.get_wm_window_hash(torch.zeros(B, 0, dtype=torch.int64), seed=42)
    assert same.tolist() == [42] * B
    per_row = torch.tensor([7, 7, 123456789, 2 ** 31 - 2, 0, 1, 2 ** 32 + 5, 99, 1000, 31337, 5, 6], dtype=torch.int64)
    out["gum_hash_same"] = same.numpy()
    out["gum_hash_rows"] = per_row.numpy()
    E.GENERATOR.manual_seed(42)
    out["gum_rs_42"] = torch.rand(V, generator=E.GENERATOR).numpy()
    cases = [("plain", 1.0, 0.0, 0), ("temp", 0.7, 0.0, 0), ("topk", 1.0, 0.0, 50), ("topp", 1.0, 0.9, 0),
             ("topp_t", 0.8, 0.5, 0), ("topk1", 1.0, 0.0, 1)]
    for hname, h in (("same", same), ("rows", per_row)):
        for name, temp, top_p, top_k in cases:
            tok = E.gumbel_sample(logits.clone(), h, use_sampling=True, temp=temp, top_p=top_p, top_k=top_k)
            out[f"gum_tok_{hname}_{name}"] = tok.numpy().astype(np.int64)
        out[f"gum_tok_{hname}_greedy"] = E.gumbel_sample(logits.clone(), h, use_sampling=False).numpy().astype(np.int64)
    out["gum_cases"] = np.array([[t, p, k] for _, t, p, k in cases], dtype=np.float64)
    out["gum_case_names"] = np.array([c[0] for c in cases])
    toks = torch.arange(B, dtype=torch.int64) * 83 % V
    out["gum_score_tokens"] = toks.numpy()
    out["gum_score_same"] = E.gumbel_score_tok(toks, same, V).numpy()
    out["gum_score_rows"] = E.gumbel_score_tok(toks, per_row, V).numpy()
    np.savez_compressed(os.path.join(HERE, "gumbel_vectors.npz"), **out)
    print("wrote gumbel_vectors.npz", os.path.getsize(os.path.join(HERE, "gumbel_vectors.npz")), "bytes")


PROD_POS = [0, 1, 31, 32, 33, 111, 112, 113, 207, 208, 209, 255]
PROD_RAR_STEPS = [0, 1, 2, 14, 15, 16, 17, 31, 32, 33, 62, 63, 64, 110, 111, 112, 113, 206, 207, 208, 209, 254, 255]


def prod_vectors(args):
    """Production-width fixtures: the kernel variants the benchmark runs -- 1536-wide GEMMs with the
    non-uniform split-K FC2 and the two-launch vocabulary head, 1 / 2 / 4 attention waves per (sequence, head), the multi-chunk
    KV loop, 6 statistics chunks -- only exist at n_embd >= 512 and cache lengths > 112 / > 208.  The reference GPT
    (mingpt.py:125-214) is run here at 2 layers x 1536 x 24 heads, block 256, 64 rows, teacher-forced through all 256 positions;
    RAR (rar.py:319-459) at 2 layers x 1280 x 16 heads (head_dim 80), 64 conditions under guidance = 128 rows, 256 steps."""
    import torch
    from deps.taming.modules.transformer.mingpt import GPT, sample_with_past
    from deps.rar.modeling.rar import RAR
    from wmar.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    from wmar_amd.utils import synth

    torch.set_num_threads(8)
    out = {}

    def ids_file(name):
        ids = []
        for line in open(os.path.join(args.ref, "assets", name)):
            ids.extend(int(t) for t in line.split(","))
        return ids

    def vq_dict(alive, vocab):
        dead = list(set(range(vocab)) - set(alive))
        return {"alive_ids": torch.tensor(alive, dtype=torch.long), "dead_ids": torch.tensor(dead, dtype=torch.long),
                "embedding": torch.zeros(vocab, 4)}

    # ------------------------------------------------------------------ Taming GPT, production width
    cfg = synth.GPTConfig(vocab_size=16384, block_size=256, n_layer=2, n_head=24, n_embd=1536)
    sd = synth.synth_gpt_state(cfg, seed=9, logit_scale=10.0, with_mask=True)
    gpt = GPT(vocab_size=cfg.vocab_size, block_size=cfg.block_size, n_layer=cfg.n_layer, n_head=cfg.n_head, n_embd=cfg.n_embd)
    gpt.load_state_dict(sd, strict=True)
    gpt.eval()
    B = 64
    rs = np.random.RandomState(20260929)
    seq = rs.randint(0, 16384, size=(B, 256)).astype(np.int64)
    seq[:, 0] = [(i * 37) % 1000 for i in range(B)]
    seq_t = torch.from_numpy(seq)
    past = None
    lgs, amax = [], []
    with torch.no_grad():
        for t in range(256):
            lg, _, present = gpt.forward_with_past(seq_t[:, t:t + 1], past=past, past_length=t)
            past = [present] if past is None else past + [present]
            lg = lg[:, -1, :]
            amax.append(lg.argmax(-1).numpy())
            if t in PROD_POS:
                lgs.append(lg[:, ::64].numpy().copy())
    out["gpt_seq"] = seq.astype(np.int16)                      # [64,256] teacher-forcing tokens
    out["gpt_pos"] = np.array(PROD_POS, dtype=np.int32)
    out["gpt_logits"] = np.stack(lgs)                           # [12,64,256]: every 64th logit
    out["gpt_argmax"] = np.stack(amax).astype(np.int16)         # [256,64]
    print("gpt teacher-forced done; |logit| max %.2f std %.2f" % (np.abs(out["gpt_logits"]).max(), out["gpt_logits"].std()), flush=True)

    wm = GentimeWatermark(vq_dict(ids_file("vqgan_alive_ids.txt"), 16384), 16384, SeedStrategy.LINEAR,
                          SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25)
    cond = torch.tensor([[7], [980], [1], [340]], dtype=torch.long)
    torch.manual_seed(11)
    toks = sample_with_past(cond, gpt, steps=256, temperature=1.0, sample_logits=True, top_k=250, top_p=0.92,
                            logit_processor=wm.spawn_logit_processor())
    out["loop_cond"] = cond.numpy()
    out["loop_tokens"] = toks.numpy().astype(np.int16)          # q: torch.manual_seed(11), 256 draws of [4,16384] Exp(1)
    out["loop_pvals"] = wm.detect(toks).numpy()
    print("gpt loop done", flush=True)

    # ------------------------------------------------------------------ RAR, production width (head_dim 80)
    class AD(dict):
        def __getattr__(self, k):
            v = self[k]
            return AD(v) if isinstance(v, dict) else v

        def get(self, k, d=None):
            return dict.get(self, k, d)

    rcfg = synth.RARConfig(hidden_size=1280, num_hidden_layers=2, num_attention_heads=16, intermediate_size=5120,
                           image_seq_len=256, codebook_size=1024, condition_num_classes=1000)
    cfgd = AD(model=dict(vq_model=dict(codebook_size=1024),
                         generator=dict(hidden_size=1280, num_hidden_layers=2, num_attention_heads=16, intermediate_size=5120,
                                        image_seq_len=256, condition_num_classes=1000, dropout=0.0, attn_drop=0.0)))
    rsd = synth.synth_rar_state(rcfg, seed=12, logit_scale=8.0)
    gen = RAR(cfgd).eval()
    gen.load_state_dict(rsd, strict=True)
    gen.set_random_ratio(0)
    wmr = GentimeWatermark(vq_dict(ids_file("rar_all_ids.txt"), 1024), 1024, SeedStrategy.LINEAR,
                           SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25)
    rec = {}
    orig = gen.forward_fn
    state = {"n": 0}

    def spy(ids, condition, **kw):
        r = orig(ids, condition, **kw)
        if state["n"] in PROD_RAR_STEPS:
            rec[state["n"]] = r[:, -1, ::8].detach().clone().numpy()
        state["n"] += 1
        return r

    gen.forward_fn = spy
    condr = torch.tensor([[(i * 13) % 1000] for i in range(64)], dtype=torch.long)
    torch.manual_seed(31)
    toks = gen.generate(condition=condr, guidance_scale=4.0, guidance_scale_pow=0.0, randomize_temperature=1.0,
                        logit_processor=wmr.spawn_logit_processor())
    out["rar_cond"] = condr.view(-1).numpy().astype(np.int16)
    out["rar_tokens"] = toks.numpy().astype(np.int16)          # [64,256]: teacher-forcing tokens for the logits below
    out["rar_steps"] = np.array(PROD_RAR_STEPS, dtype=np.int32)
    out["rar_logits"] = np.stack([rec[s] for s in PROD_RAR_STEPS])   # [23,128,128]: cond rows then uncond rows, every 8th logit
    print("rar B=64 done; |logit| max %.2f std %.2f" % (np.abs(out["rar_logits"]).max(), out["rar_logits"].std()), flush=True)
    state["n"] = -10 ** 9
    cond4 = torch.tensor([[3], [977], [0], [512]], dtype=torch.long)
    torch.manual_seed(21)
    toks4 = gen.generate(condition=cond4, guidance_scale=4.0, guidance_scale_pow=0.0, randomize_temperature=1.0,
                         logit_processor=wmr.spawn_logit_processor())
    out["rar_loop_cond"] = cond4.view(-1).numpy()
    out["rar_loop_tokens"] = toks4.numpy().astype(np.int16)    # q: manual_seed(21), rand(4,1), 256 draws of [4,1024] Exp(1)
    out["rar_loop_pvals"] = wmr.detect(toks4).numpy()
    gen.forward_fn = orig
    np.savez_compressed(os.path.join(HERE, "prod_vectors.npz"), **out)
    print("wrote prod_vectors.npz", os.path.getsize(os.path.join(HERE, "prod_vectors.npz")), "bytes")


# ------------------------------------------------------------------ bulk sampler decisions
from bulk_inputs import BULK_CHUNKS, BULK_ROWS, BULK_V, bulk_noise, bulk_rows  # noqa: E402  (same directory)


def sampler_bulk_vectors(args):
    """24 576 decisions of the reference's sampling chain (mingpt.py:348-363): GentimeWatermark._process_logits -> /T ->
    HF TopKLogitsWarper -> TopPLogitsWarper -> softmax -> torch.multinomial, V = 16384, logit scales 1 / 10 / 40, plain /
    tie-heavy / bf16-valued rows.  Only the tokens are stored; inputs are regenerated from the chunk index."""
    import torch
    import torch.nn.functional as F
    from transformers import TopKLogitsWarper, TopPLogitsWarper
    from wmar.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy

    torch.set_num_threads(8)
    ids = []
    for line in open(os.path.join(args.ref, "assets", "vqgan_alive_ids.txt")):
        ids.extend(int(t) for t in line.split(","))
    dead = list(set(range(BULK_V)) - set(ids))
    vq = {"alive_ids": torch.tensor(ids), "dead_ids": torch.tensor(dead), "embedding": torch.zeros(BULK_V, 4)}
    wm = GentimeWatermark(vq, BULK_V, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25)
    proc = wm.spawn_logit_processor()
    toks = np.zeros((BULK_CHUNKS, BULK_ROWS), dtype=np.int16)
    for c in range(BULK_CHUNKS):
        ctx, lg, top_k, top_p, T, use_wm = bulk_rows(c)
        logits = lg.clone()
        if use_wm:
            logits = proc(past_ids=ctx, logits=logits)
        logits = logits / T
        if top_k is not None:
            logits = TopKLogitsWarper(top_k=top_k)(input_ids=ctx, scores=logits)
        if top_p is not None:
            logits = TopPLogitsWarper(top_p=top_p)(input_ids=ctx, scores=logits)
        probs = F.softmax(logits, dim=-1)
        torch.manual_seed(88000 + c)
        toks[c] = torch.multinomial(probs, num_samples=1).view(-1).numpy()
        print("chunk", c, flush=True)
    np.savez_compressed(os.path.join(HERE, "sampler_bulk.npz"), tokens=toks)
    print("wrote sampler_bulk.npz", os.path.getsize(os.path.join(HERE, "sampler_bulk.npz")), "bytes")


# ------------------------------------------------------------------ harness / metrics / delta checkpoints (A13, A14, f1)
# the reduced model: wmar_amd.utils.synth.HARNESS_GPT / HARNESS_VQ (GPT 2L x 128d over the 16384-code vocabulary, VQGAN at 32 px)
HARNESS_INPUTS = [c for c in (1, 9) for _ in range(3)]          # 2 classes x 3 samples, batch 4 -> batches of 4 and 2
HARNESS_GEN = {"batch_size": 4, "temperature": 1.0, "top_k": 250, "top_p": 0.92}
HARNESS_EVAL = {"metric_names": ["pvalue", "l0", "psnr"], "augmentations": [], "max_roundtrips": 1, "orig_only": False}


def _placeholder_modules():
    """generate.py imports, at module level, wrappers and managers whose dependencies are absent here (diffusers, xformers,
    real torchvision).  None of them is on the Taming path under test (generate / fill_batch_log /
    compute_metrics_and_save_from_batch_log / compute_metric / chw_to_pillow / update_weights / TamingARMMWrapper); they are
    replaced by empty placeholder modules that only carry the imported names."""
    import types
    for name, attrs in [("wmar.augmentations.augmentation_manager", ["AugmentationManager"]),
                        ("wmar.models.chameleon_wrapper", ["ChameleonARMMWrapper"]),
                        ("wmar.models.rar_wrapper", ["RarARMMWrapper"]),
                        ("wmar.watermarking.synchronization", ["SyncManager"]),
                        ("omegaconf", ["OmegaConf"]),
                        ("deps.rar.utils.train_utils", ["create_pretrained_tokenizer"])]:
        m = types.ModuleType(name)
        for a in attrs:
            setattr(m, a, type(a, (), {}))
        sys.modules[name] = m


def harness_vectors(args):
    """The reference's generate.generate (generate.py:168-232) on a reduced Taming model: class ids + seed -> codes, p-values,
    l0, psnr, file names; compute_metric / chw_to_pillow known answers (metrics.py:20-45, utils.py:74-80);
    update_weights on encoder / decoder delta checkpoints (utils.py:47-66)."""
    import random
    import shutil
    import torch
    from PIL import Image
    _placeholder_modules()
    import generate as RG
    from deps.taming.models.cond_transformer import Net2NetTransformer
    from wmar.models.taming_wrapper import TamingARMMWrapper
    from wmar.utils.metrics import compute_metric, compute_psnr
    from wmar.utils.utils import chw_to_pillow, update_weights
    from wmar.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    from wmar_amd.utils import synth

    torch.set_num_threads(8)
    out = {}
    gcfg = synth.GPTConfig(**synth.HARNESS_GPT)
    vcfg = synth.VQConfig(**synth.HARNESS_VQ)
    gs = synth.synth_gpt_state(gcfg, seed=21, logit_scale=40.0, with_mask=True)
    vs = synth.synth_vq_state(vcfg, seed=21)
    dd = dict(double_z=False, z_channels=vcfg.z_channels, resolution=vcfg.resolution, in_channels=3, out_ch=3, ch=vcfg.ch,
              ch_mult=list(vcfg.ch_mult), num_res_blocks=vcfg.num_res_blocks, attn_resolutions=list(vcfg.attn_resolutions),
              dropout=0.0)
    net = Net2NetTransformer(
        transformer_config={"target": "deps.taming.modules.transformer.mingpt.GPT", "params": dict(synth.HARNESS_GPT)},
        first_stage_config={"target": "deps.taming.models.vqgan.VQModel",
                            "params": dict(embed_dim=vcfg.embed_dim, n_embed=vcfg.n_embed, ddconfig=dd,
                                           lossconfig={"target": "deps.taming.modules.losses.DummyLoss"})},
        cond_stage_config={"target": "deps.taming.modules.util.Labelator", "params": {"n_classes": 1000}})
    sd = {"transformer." + k: v for k, v in gs.items()}
    sd.update({"first_stage_model." + k: v for k, v in vs.items()})
    missing, unexpected = net.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.startswith(("first_stage_model.loss", "cond_stage_model")) for k in missing), (missing, unexpected)
    net.eval()

    def wrapper():
        w = object.__new__(TamingARMMWrapper)       # __init__ needs omegaconf + "cuda" + checkpoint files
        w.model = net
        w.init_alivecodes(os.path.join(args.ref, "assets", "vqgan_alive_ids.txt"))
        w.codes_size, w.image_size, w.dim_z = vcfg.codes_size, vcfg.resolution, vcfg.embed_dim
        return w

    w = wrapper()
    wm = GentimeWatermark(w.get_vq(), w.get_total_vocab_size(), SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25,
                          device="cpu")
    w.set_watermarker(wm)
    out["wm_str"] = np.array(str(wm))

    def run(chunk_id, num_chunks, tag):
        seed = 1 + 1000 * chunk_id                   # generate.py:304-308
        random.seed(seed); np.random.seed(seed); torch.manual_seed(seed)
        d = tempfile.mkdtemp(prefix="wmar_harness_")
        RG.generate(d, w, HARNESS_INPUTS, wm, HARNESS_EVAL, HARNESS_GEN, chunk_id=chunk_id, num_chunks=num_chunks)
        files = sorted(os.path.relpath(os.path.join(r, f), d) for r, _, fs in os.walk(d) for f in fs)
        out[f"{tag}_files"] = np.array(files)
        codes, metrics, pngs = [], [], []
        for f in files:
            if f.endswith(".npy"):
                codes.append(np.load(os.path.join(d, f)))
                metrics.append(json.load(open(os.path.join(d, f[:-4] + ".json"))))
                pngs.append(np.array(Image.open(os.path.join(d, f[:-4] + ".png"))))
        out[f"{tag}_codes"] = np.stack(codes).astype(np.int64)              # in the order of the sorted .npy names
        out[f"{tag}_pvalue"] = np.array([m["pvalue"] for m in metrics], dtype=np.float64)
        out[f"{tag}_l0"] = np.array([m["l0"] for m in metrics], dtype=np.float64)
        out[f"{tag}_psnr"] = np.array([m["psnr"] for m in metrics], dtype=np.float64)
        out[f"{tag}_png"] = np.stack(pngs)
        if tag == "job":
            # f3: the reference's own result-directory reader and its TPR rule (wmar/utils/analyzer.py:186-238; the rule
            # `np.sum(np.array(pvals) < 0.01) / len(pvals)` is inline in plot_robustness :376-381, :419-424)
            from wmar.utils.analyzer import Analyzer
            _, am, paths, N = Analyzer.get_metrics_imagepaths_N([("roundtrips", None, [0, 1])], "lbl", os.path.dirname(d),
                                                                 os.path.basename(d), str(wm))
            out["an_N"] = np.array(N)
            out["an_keys"] = np.array(sorted(am))
            for k in sorted(am):
                pv = [m["pvalue"] for m in am[k]]
                out[f"an_{k}_pvalues_sorted"] = np.array(sorted(pv), dtype=np.float64)
                out[f"an_{k}_l0_sorted"] = np.array(sorted(m["l0"] for m in am[k]), dtype=np.float64)
                for thr in (0.01, 0.25):
                    out[f"an_{k}_tpr_at_{thr}"] = np.array(np.sum(np.array(pv) < thr) / len(pv))
            out["an_orig_image_classes"] = np.array(sorted(paths))
            out["an_orig_images_per_class"] = np.array([len(paths[c]) for c in sorted(paths)])
        shutil.rmtree(d)

    run(0, 1, "job")
    run(0, 2, "chunk0of2")
    run(1, 2, "chunk1of2")

    # ---- A13 known answers on crafted inputs
    rs = np.random.RandomState(5)
    a = rs.randint(0, 16384, size=64).astype(np.int64)
    b = a.copy()
    b[rs.choice(64, 11, replace=False)] += 1
    ia = (rs.rand(3, 32, 32).astype(np.float32) * 2.4 - 1.2)                  # outside [-1,1] too: the clip matters
    ib = np.clip(ia + rs.randn(3, 32, 32).astype(np.float32) * 0.05, -1.5, 1.5)
    ticks = (np.arange(256, dtype=np.float64) + 0.5) / 255.0 * 2.0 - 1.0      # values that land on k + 0.5 before rounding
    ia[0, 0, :] = ticks[:32].astype(np.float32)
    ia[0, 1, :] = ticks[100:132].astype(np.float32)
    pa, pb = chw_to_pillow(ia), chw_to_pillow(torch.from_numpy(ib))
    out["met_code_a"], out["met_code_b"] = a, b
    out["met_img_a"], out["met_img_b"] = ia, ib
    out["met_u8_a"], out["met_u8_b"] = np.array(pa), np.array(pb)
    out["met_l0"] = np.array(compute_metric("l0", b, a, pb, pa, wm, "roundtrips", 0))
    out["met_psnr"] = np.array(compute_metric("psnr", b, a, pb, pa, wm, "roundtrips", 0))
    out["met_psnr_same"] = np.array(compute_psnr(pa, pa))
    out["met_pvalue"] = np.array(compute_metric("pvalue", b, a, pb, pa, wm, "roundtrips", 0))
    out["met_bpp"] = np.array(-1.0 if compute_metric("bpp", b, a, pb, pa, wm, "roundtrips", 0) is None else 1.0)

    # ---- f1: delta checkpoints through the reference's update_weights
    codes = torch.from_numpy(rs.randint(0, 16384, size=(2, vcfg.codes_size ** 2)).astype(np.int64))
    img0 = w.codes_to_images(codes)
    c0 = w.images_to_codes(img0)
    tmpd = tempfile.mkdtemp(prefix="wmar_delta_")
    torch.save(synth.synth_delta(vs, "decoder.", seed=1), os.path.join(tmpd, "dec_delta.pth"))
    torch.save({"state_dict": synth.synth_delta(vs, "encoder.", seed=2)}, os.path.join(tmpd, "enc_delta.pth"))
    update_weights(w.get_image_tokenizer().decoder, os.path.join(tmpd, "dec_delta.pth"))
    img1 = w.codes_to_images(codes)
    update_weights(w.get_image_tokenizer().encoder, os.path.join(tmpd, "enc_delta.pth"))
    c1 = w.images_to_codes(img1)
    shutil.rmtree(tmpd)
    out["delta_codes"] = codes.numpy()
    out["delta_img_before"], out["delta_img_after"] = img0.numpy(), img1.numpy()
    out["delta_codes_before"], out["delta_codes_after"] = c0.numpy(), c1.numpy()
    print("delta: max|dimg| %.4f, codes changed by the encoder patch: %d / %d" % (
        float((img1 - img0).abs().max()), int((w.images_to_codes(img1) != c1).sum()), c1.numel()))
    np.savez_compressed(os.path.join(HERE, "harness_vectors.npz"), **out)
    print("wrote harness_vectors.npz", os.path.getsize(os.path.join(HERE, "harness_vectors.npz")), "bytes")
    for k in ("job_files", "job_pvalue", "job_l0", "job_psnr"):
        print(k, out[k])


def spatial_vectors(args):
    """f4: the reference's sampling loop (mingpt.py:326-368) under SeedStrategy.SPATIAL (gentime_watermark.py:242-263): h = 1
    (context = the token above at row starts, else the previous token) and h = 3 (up-left, up, left), spatial_dim 4, 16 steps
    of the 2-layer / 128-wide GPT of the other loop fixtures.  Exercises the SPATIAL branch of the fused sampler inside the
    captured generation step."""
    import torch
    from deps.taming.modules.transformer.mingpt import GPT, sample_with_past
    from wmar.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    from wmar_amd.utils import synth
    import contextlib, io

    ids = []
    for line in open(os.path.join(args.ref, "assets", "vqgan_alive_ids.txt")):
        ids.extend(int(t) for t in line.split(","))
    dead = list(set(range(16384)) - set(ids))
    vq = {"alive_ids": torch.tensor(ids), "dead_ids": torch.tensor(dead), "embedding": torch.zeros(16384, 4)}
    cfg = synth.GPTConfig(vocab_size=16384, block_size=16, n_layer=2, n_head=4, n_embd=128)
    sd = synth.synth_gpt_state(cfg, seed=3, logit_scale=40.0, with_mask=True)
    gpt = GPT(vocab_size=cfg.vocab_size, block_size=cfg.block_size, n_layer=cfg.n_layer, n_head=cfg.n_head, n_embd=cfg.n_embd)
    gpt.load_state_dict(sd, strict=True)
    gpt.eval()
    cond = torch.tensor([[7], [980], [1], [340]], dtype=torch.long)
    out = {"cond": cond.numpy()}
    for h in (1, 3):
        wm = GentimeWatermark(vq, 16384, SeedStrategy.SPATIAL, SplitStrategy.RANDOM_STRATIFIED, h, 2.0, 0.25, spatial_dim=4)
        torch.manual_seed(11)
        with contextlib.redirect_stdout(io.StringIO()):
            toks = sample_with_past(cond, gpt, steps=16, temperature=1.0, sample_logits=True, top_k=250, top_p=0.92,
                                    logit_processor=wm.spawn_logit_processor())
            pv, masks = wm.detect(toks, return_masks=True)
        out[f"tokens_h{h}"] = toks.numpy()
        out[f"pvals_h{h}"] = pv.numpy()
        out[f"masks_h{h}"] = np.array(masks, dtype=np.int8)
    np.savez_compressed(os.path.join(HERE, "spatial_vectors.npz"), **out)
    print("wrote spatial_vectors.npz", os.path.getsize(os.path.join(HERE, "spatial_vectors.npz")), "bytes")


def linear_h_vectors(args):
    """LINEAR seeding with context sizes beyond 3 (gentime_watermark.py:236-241 takes any `context_size`): the reference's sampling
    loop (mingpt.py:326-368) with h = 4 and h = 5 on the 2-layer / 128-wide GPT of the other loop fixtures (V = 16384, 24 steps),
    its detector (p-values, masks) on those tokens, and the logit processor on a past that is too short for some rows."""
    import torch
    from deps.taming.modules.transformer.mingpt import GPT, sample_with_past
    from wmar.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    from wmar_amd.utils import synth
    import contextlib, io

    ids = []
    for line in open(os.path.join(args.ref, "assets", "vqgan_alive_ids.txt")):
        ids.extend(int(t) for t in line.split(","))
    dead = list(set(range(16384)) - set(ids))
    vq = {"alive_ids": torch.tensor(ids), "dead_ids": torch.tensor(dead), "embedding": torch.zeros(16384, 4)}
    cfg = synth.GPTConfig(vocab_size=16384, block_size=24, n_layer=2, n_head=4, n_embd=128)
    sd = synth.synth_gpt_state(cfg, seed=3, logit_scale=40.0, with_mask=True)
    gpt = GPT(vocab_size=cfg.vocab_size, block_size=cfg.block_size, n_layer=cfg.n_layer, n_head=cfg.n_head, n_embd=cfg.n_embd)
    gpt.load_state_dict(sd, strict=True)
    gpt.eval()
    cond = torch.tensor([[7], [980], [1], [340]], dtype=torch.long)
    out = {"cond": cond.numpy()}
    rs = np.random.RandomState(44)
    logits = torch.from_numpy(rs.randn(3, 16384).astype(np.float32) * 3)
    for h in (4, 5):
        wm = GentimeWatermark(vq, 16384, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, h, 2.0, 0.25)
        torch.manual_seed(11)
        with contextlib.redirect_stdout(io.StringIO()):
            toks = sample_with_past(cond, gpt, steps=24, temperature=1.0, sample_logits=True, top_k=250, top_p=0.92,
                                    logit_processor=wm.spawn_logit_processor())
            pv, masks = wm.detect(toks, return_masks=True)
        out[f"tokens_h{h}"] = toks.numpy()
        out[f"pvals_h{h}"] = pv.numpy()
        out[f"masks_h{h}"] = np.array(masks, dtype=np.int8)
        # logit processor: a past of exactly h tokens (bias applied) and one of h - 1 (every row skipped, gentime_watermark.py:237-240)
        past = torch.from_numpy(rs.randint(0, 16384, size=(3, h)).astype(np.int64))
        out[f"proc_past_h{h}"] = past.numpy()
        out[f"proc_out_h{h}"] = wm.spawn_logit_processor()(past_ids=past, logits=logits.clone()).numpy()
        out[f"proc_short_out_h{h}"] = wm.spawn_logit_processor()(past_ids=past[:, :h - 1], logits=logits.clone()).numpy()
    out["proc_logits"] = logits.numpy()
    np.savez_compressed(os.path.join(HERE, "linear_h_vectors.npz"), **out)
    print("wrote linear_h_vectors.npz", os.path.getsize(os.path.join(HERE, "linear_h_vectors.npz")), "bytes")


def fullsize_vq_vectors(args):
    """The three image tokenizers at the shapes the benchmarks run, from the reference's own
    modules on seeded weights (synth.*_state(seed) regenerates them bit for bit wherever the fixture is used):
      * Taming VQGAN f16/16384: ch 128, mult (1,1,2,2,4), 2 res blocks, attention at 16, 256 x 256, 16384 x 256 codebook
        (deps/taming/modules/diffusionmodules/model.py:407-434, 507-538; deps/taming/modules/vqvae/quantize.py:272-331);
      * MaskGIT-VQGAN of RAR at 256 x 256, 1024 codes (deps/rar/modeling/modules/maskgit_vqgan.py:157-362);
      * Chameleon's VQGAN at 512 x 512, 8192 codes (deps/chameleon/inference/vqgan.py:330-570).
    Stored per model: the codes, every 4th pixel of the decoded images (a different phase per image), the pre-quantisation vectors of
    64 positions, the re-encoded codes with the reference's margin between the best and the second-best code (a code whose margin is
    below the fp32 noise of the distance may legitimately differ), and for Taming the detector's p-values of both code sets."""
    import torch
    from deps.taming.modules.diffusionmodules.model import Decoder, Encoder
    from deps.taming.modules.vqvae.quantize import VectorQuantizer2
    from wmar.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    from wmar_amd.utils import synth

    torch.set_num_threads(8)
    out = {}
    rs = np.random.RandomState(77)

    def margins(z_flat, emb):
        # quantize.py:279-283: d = sum z^2 + sum e^2 - 2 z e^T ; the two smallest distances per row
        d = torch.sum(z_flat ** 2, dim=1, keepdim=True) + torch.sum(emb ** 2, dim=1) - 2 * torch.einsum("bd,dn->bn", z_flat, emb.t())
        two = torch.topk(d, 2, dim=1, largest=False).values
        return (two[:, 1] - two[:, 0]).numpy()

    def sub(img):        # every 4th pixel, phase (b, 2b+1) for image b
        return np.stack([img[b, :, (b % 4)::4, ((2 * b + 1) % 4)::4] for b in range(img.shape[0])])

    # ---- Taming
    vcfg = synth.TAMING_VQ
    vsd = synth.synth_vq_state(vcfg, seed=31)
    dd = dict(double_z=False, z_channels=vcfg.z_channels, resolution=vcfg.resolution, in_channels=3, out_ch=3, ch=vcfg.ch,
              ch_mult=list(vcfg.ch_mult), num_res_blocks=vcfg.num_res_blocks, attn_resolutions=list(vcfg.attn_resolutions), dropout=0.0)
    enc, dec = Encoder(**dd).eval(), Decoder(**dd).eval()
    enc.load_state_dict({k[len("encoder."):]: v for k, v in vsd.items() if k.startswith("encoder.")}, strict=True)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in vsd.items() if k.startswith("decoder.")}, strict=True)
    vq = VectorQuantizer2(vcfg.n_embed, vcfg.embed_dim, beta=0.25).eval()
    vq.load_state_dict({"embedding.weight": vsd["quantize.embedding.weight"]}, strict=True)
    qc = torch.nn.Conv2d(vcfg.z_channels, vcfg.embed_dim, 1)
    pqc = torch.nn.Conv2d(vcfg.embed_dim, vcfg.z_channels, 1)
    qc.load_state_dict({"weight": vsd["quant_conv.weight"], "bias": vsd["quant_conv.bias"]})
    pqc.load_state_dict({"weight": vsd["post_quant_conv.weight"], "bias": vsd["post_quant_conv.bias"]})
    S = vcfg.codes_size
    codes = torch.from_numpy(rs.randint(0, vcfg.n_embed, size=(2, S * S)).astype(np.int64))
    with torch.no_grad():
        zq = vq.get_codebook_entry(codes.reshape(-1), shape=(2, S, S, vcfg.embed_dim))
        img = dec(pqc(zq)).clamp(-1, 1)                                   # cond_transformer.py:186-192, taming_wrapper.py:79-84
        h = qc(enc(img))
        _, _, info = vq(h)
        codes2 = info[2].view(2, -1)
        zf = h.permute(0, 2, 3, 1).reshape(-1, vcfg.embed_dim)
        out["tam_margin"] = margins(zf, vq.embedding.weight)
    out["tam_codes"] = codes.numpy()
    out["tam_pixels"] = sub(img.numpy())
    out["tam_prequant"] = zf.numpy()[::8]
    out["tam_codes_roundtrip"] = codes2.numpy()

    def ids_file(name):
        ids = []
        for line in open(os.path.join(args.ref, "assets", name)):
            ids.extend(int(t) for t in line.split(","))
        return ids
    alive = ids_file("vqgan_alive_ids.txt")
    dead = list(set(range(16384)) - set(alive))
    wm = GentimeWatermark({"alive_ids": torch.tensor(alive), "dead_ids": torch.tensor(dead), "embedding": torch.zeros(16384, 4)}, 16384,
                          SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25)
    out["tam_pvals"] = wm.detect(codes).numpy()
    out["tam_pvals_roundtrip"] = wm.detect(codes2).numpy()
    print("taming full size: roundtrip l0 %.3f, min margin %.3e" % (float((codes2 != codes).float().mean()), float(out["tam_margin"].min())))

    # ---- MaskGIT-VQGAN (RAR)
    from deps.rar.modeling.modules.maskgit_vqgan import Decoder as MDecoder, Encoder as MEncoder, VectorQuantizer as MVQ

    class AD(dict):
        def __getattr__(self, k):
            return self[k]

        def get(self, k, d=None):
            return dict.get(self, k, d)
    mcfg = synth.MASKGIT_VQ
    msd = synth.synth_maskgit_state(mcfg, seed=33)
    tc = AD(channel_mult=list(mcfg.channel_mult), num_resolutions=mcfg.num_resolutions, dropout=0.0, hidden_channels=mcfg.hidden_channels,
            num_channels=3, num_res_blocks=mcfg.num_res_blocks, resolution=mcfg.resolution, z_channels=mcfg.z_channels)
    menc, mdec = MEncoder(tc).eval(), MDecoder(tc).eval()
    mvq = MVQ(mcfg.num_embeddings, mcfg.z_channels, 0.25).eval()
    menc.load_state_dict({k[8:]: v for k, v in msd.items() if k.startswith("encoder.")}, strict=True)
    mdec.load_state_dict({k[8:]: v for k, v in msd.items() if k.startswith("decoder.")}, strict=True)
    mvq.load_state_dict({"embedding.weight": msd["quantize.embedding.weight"]}, strict=True)
    S = mcfg.codes_size
    codes = torch.from_numpy(rs.randint(0, mcfg.num_embeddings, size=(2, S * S)).astype(np.int64))
    with torch.no_grad():
        img01 = torch.clamp(mdec(mvq.get_codebook_entry(codes)), 0.0, 1.0)          # titok.py:81-85
        img = torch.clamp(img01 * 2.0 - 1.0, -1.0, 1.0)                              # rar_wrapper.py:112-114
        h = menc((img + 1.0) / 2.0)                                                  # rar_wrapper.py:124-125
        _, idx, _ = mvq(h)
        zf = h.permute(0, 2, 3, 1).reshape(-1, mcfg.z_channels)
        out["mg_margin"] = margins(zf, mvq.embedding.weight)
    out["mg_codes"] = codes.numpy()
    out["mg_pixels"] = sub(img.numpy())
    out["mg_prequant"] = zf.numpy()[::8]
    out["mg_codes_roundtrip"] = idx.view(2, -1).numpy().astype(np.int64)
    print("maskgit full size: roundtrip l0 %.3f, min margin %.3e" % (float((idx.view(2, -1) != codes).float().mean()), float(out["mg_margin"].min())))

    # ---- Chameleon VQGAN at 512 x 512 (one image)
    from deps.chameleon.inference.vqgan import VQModel
    ccfg = synth.CHAMELEON_VQ
    csd = synth.synth_vq_state(ccfg, seed=35)
    cvq = VQModel(ddconfig=dict(double_z=False, z_channels=ccfg.z_channels, resolution=ccfg.resolution, in_channels=3, out_ch=3, ch=ccfg.ch,
                                ch_mult=list(ccfg.ch_mult), num_res_blocks=ccfg.num_res_blocks, attn_resolutions=[], dropout=0.0),
                  n_embed=ccfg.n_embed, embed_dim=ccfg.embed_dim).eval()
    cvq.load_state_dict(csd, strict=True)
    S = ccfg.codes_size
    codes = torch.from_numpy(rs.randint(0, ccfg.n_embed, size=(1, S * S)).astype(np.int64))
    with torch.no_grad():
        zq = cvq.quantize.get_codebook_entry(codes.view(-1), (1, S, S, ccfg.embed_dim))
        img = cvq.decode(zq)                                                         # image_tokenizer.py:118-126 (no clamp in the model)
        h = cvq.quant_conv(cvq.encoder(img))
        _, _, (_, _, idx) = cvq.encode(img)
        zf = h.permute(0, 2, 3, 1).reshape(-1, ccfg.embed_dim)
        out["ch_margin"] = margins(zf, cvq.quantize.embedding.weight)
    out["ch_codes"] = codes.numpy()
    out["ch_pixels"] = sub(img.numpy())
    out["ch_prequant"] = zf.numpy()[::16]
    out["ch_codes_roundtrip"] = idx.view(1, -1).numpy().astype(np.int64)
    print("chameleon full size: roundtrip l0 %.3f, min margin %.3e" % (float((idx.view(1, -1) != codes).float().mean()), float(out["ch_margin"].min())))
    np.savez_compressed(os.path.join(HERE, "fullsize_vq_vectors.npz"), **out)
    print("wrote fullsize_vq_vectors.npz", os.path.getsize(os.path.join(HERE, "fullsize_vq_vectors.npz")), "bytes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--only", default="", help="'gumbel' / 'chameleon' / 'prod' / 'sampler' / 'harness' / 'spatial' / 'linearh' / 'fullvq': regenerate only that fixture file")
    args = ap.parse_args()
    if args.only == "gumbel":
        gumbel_vectors(args)
        return
    if args.only in ("chameleon", "prod", "sampler", "harness", "spatial", "linearh", "fullvq"):
        tmp = tempfile.mkdtemp(prefix="wmar_stubs_")
        _stubs(tmp)
        sys.path[:0] = [tmp, args.ref, REPO]
        os.chdir(args.ref)
        {"chameleon": chameleon_vectors, "prod": prod_vectors, "sampler": sampler_bulk_vectors, "harness": harness_vectors, "spatial": spatial_vectors,
         "linearh": linear_h_vectors, "fullvq": fullsize_vq_vectors}[args.only](args)
        return
    tmp = tempfile.mkdtemp(prefix="wmar_stubs_")
    _stubs(tmp)
    sys.path[:0] = [tmp, args.ref, REPO]
    os.chdir(args.ref)

    import torch
    from deps.taming.modules.diffusionmodules.model import Decoder, Encoder
    from deps.taming.modules.transformer.mingpt import GPT, sample_with_past
    from deps.taming.modules.vqvae.quantize import VectorQuantizer2
    from wmar.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    from scipy import special

    from wmar_amd.utils import synth

    torch.set_num_threads(8)
    out = {}

    def ids_file(name):
        ids = []
        for line in open(os.path.join(args.ref, "assets", name)):
            ids.extend(int(t) for t in line.split(","))
        return ids

    def vq_dict(alive, vocab):
        # armm_wrapper.py:42-55: dead = list(set(range(V)) - set(alive))
        dead = list(set(range(vocab)) - set(alive))
        return {"alive_ids": torch.tensor(alive, dtype=torch.long), "dead_ids": torch.tensor(dead, dtype=torch.long),
                "embedding": torch.zeros(vocab, 4)}, dead

    # ---------------------------------------------------------------- 1. key KATs
    kat = {"randperm": [], "keys": {}}
    for n, seed in [(971, 0), (15413, 12345), (57344, 7), (1, 3), (2, 3), (16384, 2**32 + 5), (1024, 15485863 * 77)]:
        g = torch.Generator().manual_seed(seed)
        p = torch.randperm(n, generator=g)
        kat["randperm"].append({"n": n, "seed": seed, "first": p[:8].tolist(), "last": p[-4:].tolist(),
                                "sha256": hashlib.sha256(p.numpy().astype("<i8").tobytes()).hexdigest()})
    key_cfgs = {
        "taming": dict(alive="vqgan_alive_ids.txt", vocab=16384, seed="linear", split="stratifiedrand", h=1, gamma=0.25),
        "taming_rand": dict(alive="vqgan_alive_ids.txt", vocab=16384, seed="linear", split="rand", h=1, gamma=0.25),
        "taming_h2_g50": dict(alive="vqgan_alive_ids.txt", vocab=16384, seed="linear", split="stratifiedrand", h=2, gamma=0.5),
        "rar": dict(alive="rar_all_ids.txt", vocab=1024, seed="linear", split="stratifiedrand", h=1, gamma=0.25),
        "chameleon_fixed": dict(alive="chameleon_all_ids.txt", vocab=65536, seed="fixed", split="stratifiedrand", h=0, gamma=0.25),
    }
    wms = {}
    for name, c in key_cfgs.items():
        alive = ids_file(c["alive"])
        vq, dead = vq_dict(alive, c["vocab"])
        wm = GentimeWatermark(vq, c["vocab"], SeedStrategy(c["seed"]), SplitStrategy(c["split"]), c["h"], 2.0, c["gamma"])
        wms[name] = wm
        entry = dict(c)
        entry["dead_sha256"] = hashlib.sha256(np.array(dead, dtype="<i8").tobytes()).hexdigest()
        entry["dead_is_ascending"] = bool(dead == sorted(dead))
        entry["str"] = str(wm)
        ctxs = [[0] * c["h"], [5] * c["h"], [c["vocab"] - 1] * c["h"], [999] + [3] * (c["h"] - 1) if c["h"] else []]
        entry["contexts"] = []
        for ctx in ctxs:
            gl = wm._get_greenlist_ids_for_context(torch.tensor(ctx, dtype=torch.long))
            entry["contexts"].append({"ctx": ctx, "len": len(gl), "first": gl[:8].tolist(), "last": gl[-6:].tolist(),
                                      "sha256": hashlib.sha256(gl.numpy().astype("<i8").tobytes()).hexdigest()})
        if c["seed"] != "fixed":
            h = hashlib.sha256()
            for s in range(16):
                gl = wm._get_greenlist_ids_for_context(torch.tensor([s] + [0] * (c["h"] - 1), dtype=torch.long))
                h.update(np.sort(gl.numpy()).astype("<i4").tobytes())
            entry["sha256_sorted_ctx0_15"] = h.hexdigest()
        kat["keys"][name] = entry
    kat["betainc"] = [{"a": a, "b": b, "x": x, "v": float(special.betainc(a, b, x))}
                      for a, b, x in [(0, 256, .25), (1, 255, .25), (64, 192, .25), (100, 156, .25), (255, 1, .25),
                                      (30, 226, .25), (128, 128, .5), (200, 56, .5), (1, 1, .25), (3, 1021, .1),
                                      (500, 524, .25), (250, 6, 0.25), (70, 186, 0.25)]]
    json.dump(kat, open(os.path.join(HERE, "key_kat.json"), "w"), indent=1)

    # ------------------------------------------------------- 2. logit processor (A4)
    rs = np.random.RandomState(1234)
    wm = wms["taming"]
    past = torch.from_numpy(rs.randint(0, 16384, size=(4, 5)).astype(np.int64))
    past[0, -1] = 7  # a class-id-like context
    logits = torch.from_numpy(rs.randn(4, 16384).astype(np.float32))
    proc = wm.spawn_logit_processor()
    lg_out = proc(past_ids=past, logits=logits.clone())
    out["proc_taming_past"] = past.numpy()
    out["proc_taming_logits"] = logits.numpy()
    out["proc_taming_changed_sha256"] = np.frombuffer(hashlib.sha256(
        np.ascontiguousarray((lg_out != logits).numpy()).tobytes()).digest(), dtype=np.uint8)
    out["proc_taming_out"] = lg_out.numpy()
    # h=2: a row with t < h is skipped (ValueError swallowed)
    wm2 = wms["taming_h2_g50"]
    wm2.delta = 1.5
    past2 = torch.from_numpy(rs.randint(0, 16384, size=(2, 1)).astype(np.int64))
    out["proc_h2_short_out"] = wm2.spawn_logit_processor()(past_ids=past2, logits=logits[:2].clone()).numpy()
    out["proc_h2_short_past"] = past2.numpy()
    past3 = torch.from_numpy(rs.randint(0, 16384, size=(2, 3)).astype(np.int64))
    out["proc_h2_past"] = past3.numpy()
    out["proc_h2_out"] = wm2.spawn_logit_processor()(past_ids=past3, logits=logits[:2].clone()).numpy()
    # spatial h=3 / h=1
    vq, _ = vq_dict(ids_file("vqgan_alive_ids.txt"), 16384)
    wms3 = GentimeWatermark(vq, 16384, SeedStrategy.SPATIAL, SplitStrategy.RANDOM_STRATIFIED, 3, 2.0, 0.25, spatial_dim=4)
    past4 = torch.from_numpy(rs.randint(0, 16384, size=(2, 6)).astype(np.int64))
    out["proc_sp3_past"] = past4.numpy()
    out["proc_sp3_out"] = wms3.spawn_logit_processor()(past_ids=past4, logits=logits[:2].clone()).numpy()
    import contextlib, io
    wms1 = GentimeWatermark(vq, 16384, SeedStrategy.SPATIAL, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, spatial_dim=4)
    with contextlib.redirect_stdout(io.StringIO()):
        past5 = torch.from_numpy(rs.randint(0, 16384, size=(2, 8)).astype(np.int64))   # t % S == 0 -> [-S]
        out["proc_sp1a_past"] = past5.numpy()
        out["proc_sp1a_out"] = wms1.spawn_logit_processor()(past_ids=past5, logits=logits[:2].clone()).numpy()
        past6 = torch.from_numpy(rs.randint(0, 16384, size=(2, 7)).astype(np.int64))   # else [-1]
        out["proc_sp1b_past"] = past6.numpy()
        out["proc_sp1b_out"] = wms1.spawn_logit_processor()(past_ids=past6, logits=logits[:2].clone()).numpy()

    # ------------------------------------------- 3. whole sampling loop, small GPT (A5-A8)
    cfg = synth.GPTConfig(vocab_size=16384, block_size=16, n_layer=2, n_head=4, n_embd=128)
    sd = synth.synth_gpt_state(cfg, seed=3, logit_scale=40.0, with_mask=True)
    gpt = GPT(vocab_size=cfg.vocab_size, block_size=cfg.block_size, n_layer=cfg.n_layer, n_head=cfg.n_head,
              n_embd=cfg.n_embd)
    gpt.load_state_dict(sd, strict=True)  # pins the checkpoint key layout
    gpt.eval()
    wm = wms["taming"]
    wm.delta = 2.0
    cond = torch.tensor([[7], [980], [1], [340]], dtype=torch.long)
    rec = {"logits": [], "q": []}
    orig_fwd = gpt.forward_with_past

    def spy(idx, **kw):
        r = orig_fwd(idx, **kw)
        rec["logits"].append(r[0][:, -1, :].detach().clone().numpy())
        return r

    gpt.forward_with_past = spy
    for tag, (tk, tp, T) in {"k250p92": (250, 0.92, 1.0), "k100p80T13": (100, 0.8, 1.3), "nok_p95": (None, 0.95, 0.9),
                             "k50nop": (50, None, 1.0), "plain": (None, None, 1.0)}.items():
        rec["logits"].clear()
        torch.manual_seed(11)
        toks = sample_with_past(cond, gpt, steps=16, temperature=T, sample_logits=True, top_k=tk, top_p=tp,
                                logit_processor=wm.spawn_logit_processor())
        torch.manual_seed(11)  # the noise multinomial drew, step by step
        q = np.stack([torch.empty(4, 16384).exponential_(1).numpy() for _ in range(16)])
        out[f"loop_{tag}_tokens"] = toks.numpy()
        if tag == "k250p92":
            out["loop_logits"] = np.stack(rec["logits"])[:6]      # [6,4,V] raw model logits (steps 0..5)
            out["loop_q"] = q[:6]
    out["loop_cond"] = cond.numpy()
    # unwatermarked run
    torch.manual_seed(11)
    out["loop_nowm_tokens"] = sample_with_past(cond, gpt, steps=16, temperature=1.0, sample_logits=True, top_k=250,
                                               top_p=0.92, logit_processor=None).numpy()
    gpt.forward_with_past = orig_fwd

    # GPT step fixture: logits at steps 0..3 for a fixed token sequence (teacher forced)
    seq = torch.from_numpy(rs.randint(0, 16384, size=(3, 6)).astype(np.int64))
    past = None
    lgs = []
    for t in range(6):
        lg, _, present = gpt.forward_with_past(seq[:, t:t + 1], past=past, past_length=t)
        past = [present] if past is None else past + [present]
        lgs.append(lg[:, -1, :].detach().numpy())
    out["gpt_seq"] = seq.numpy()
    out["gpt_logits"] = np.stack(lgs)[:, :, ::16].copy()  # every 16th logit, [6,3,1024]
    out["gpt_logits_argmax"] = np.stack(lgs).argmax(-1)

    # ------------------------------------------------------------------ 4. detector
    wm = wms["taming"]
    torch.manual_seed(123)
    codes = torch.randint(0, 16384, (2, 256))
    out["det_codes_rand"] = codes.numpy()
    out["det_pvals_rand"] = wm.detect(codes).numpy()
    gen_codes = torch.from_numpy(out["loop_k250p92_tokens"])  # 16 watermarked tokens per row
    pv, masks = wm.detect(gen_codes, return_masks=True)
    out["det_pvals_gen"] = pv.numpy()
    out["det_masks_gen"] = np.array(masks, dtype=np.int8)
    rep = torch.tensor([[5, 9, 5, 9, 5, 9, 5, 9, 11, 5, 9, 3]], dtype=torch.long)  # repeated bigrams
    pv, masks = wm.detect(rep, return_masks=True)
    out["det_codes_rep"] = rep.numpy()
    out["det_pvals_rep"] = pv.numpy()
    out["det_masks_rep"] = np.array(masks, dtype=np.int8)
    pv, masks = wms["taming_h2_g50"].detect(codes[:, :40], return_masks=True)
    out["det_pvals_h2"] = pv.numpy()
    out["det_masks_h2"] = np.array(masks, dtype=np.int8)
    out["det_pvals_rar"] = wms["rar"].detect((torch.arange(256) % 1024).view(1, -1)).numpy()
    cham_codes = torch.from_numpy(rs.randint(0, 65536, size=(1, 64)).astype(np.int64))
    out["det_codes_cham"] = cham_codes.numpy()
    out["det_pvals_cham"] = wms["chameleon_fixed"].detect(cham_codes).numpy()
    sp_codes = torch.from_numpy(rs.randint(0, 16384, size=(2, 16)).astype(np.int64))
    out["det_codes_sp"] = sp_codes.numpy()
    pv, masks = wms3.detect(sp_codes, return_masks=True)
    out["det_pvals_sp3"] = pv.numpy()
    out["det_masks_sp3"] = np.array(masks, dtype=np.int8)
    pv, masks = wms1.detect(sp_codes, return_masks=True)
    out["det_pvals_sp1"] = pv.numpy()
    out["det_masks_sp1"] = np.array(masks, dtype=np.int8)

    # -------------------------------------------------------------------- 5. VQGAN
    vcfg = synth.VQConfig(ch=32, ch_mult=(1, 2, 2), num_res_blocks=1, attn_resolutions=(8,), resolution=32,
                          z_channels=16, embed_dim=8, n_embed=512)
    vsd = synth.synth_vq_state(vcfg, seed=5)
    dd = dict(double_z=False, z_channels=vcfg.z_channels, resolution=vcfg.resolution, in_channels=3, out_ch=3,
              ch=vcfg.ch, ch_mult=list(vcfg.ch_mult), num_res_blocks=vcfg.num_res_blocks,
              attn_resolutions=list(vcfg.attn_resolutions), dropout=0.0)
    enc, dec = Encoder(**dd).eval(), Decoder(**dd).eval()
    enc.load_state_dict({k[len("encoder."):]: v for k, v in vsd.items() if k.startswith("encoder.")}, strict=True)
    dec.load_state_dict({k[len("decoder."):]: v for k, v in vsd.items() if k.startswith("decoder.")}, strict=True)
    vq = VectorQuantizer2(vcfg.n_embed, vcfg.embed_dim, beta=0.25).eval()
    vq.load_state_dict({"embedding.weight": vsd["quantize.embedding.weight"]}, strict=True)
    qc = torch.nn.Conv2d(vcfg.z_channels, vcfg.embed_dim, 1)
    pqc = torch.nn.Conv2d(vcfg.embed_dim, vcfg.z_channels, 1)
    qc.load_state_dict({"weight": vsd["quant_conv.weight"], "bias": vsd["quant_conv.bias"]})
    pqc.load_state_dict({"weight": vsd["post_quant_conv.weight"], "bias": vsd["post_quant_conv.bias"]})
    S = vcfg.codes_size
    vcodes = torch.from_numpy(rs.randint(0, vcfg.n_embed, size=(2, S * S)).astype(np.int64))
    with torch.no_grad():
        zq = vq.get_codebook_entry(vcodes.reshape(-1), shape=(2, S, S, vcfg.embed_dim))
        img = dec(pqc(zq)).clamp(-1, 1)
        h = qc(enc(img))
        _, _, info = vq(h)
        codes2 = info[2].view(2, -1)
    out["vq_codes"] = vcodes.numpy()
    out["vq_images"] = img.numpy()
    out["vq_prequant"] = h.permute(0, 2, 3, 1).reshape(-1, vcfg.embed_dim).numpy()
    out["vq_codes_roundtrip"] = codes2.numpy()

    rar_vectors(args, synth, wms, rs)
    gumbel_vectors(args)
    chameleon_vectors(args)
    prod_vectors(args)
    sampler_bulk_vectors(args)
    np.savez_compressed(os.path.join(HERE, "reference_vectors.npz"), **out)
    sz = os.path.getsize(os.path.join(HERE, "reference_vectors.npz"))
    print("wrote reference_vectors.npz", sz, "bytes; key_kat.json")
    spatial_vectors(args)
    fullsize_vq_vectors(args)
    harness_vectors(args)      # last: it plants placeholder modules for generate.py's unrelated imports


if __name__ == "__main__":
    main()
