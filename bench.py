"""Headline benchmark: watermarked images/sec at 256x256 (16x16 tokens), batch 64 per GPU.

One "step" = one pass of the hot path over one batch of synthetic class labels:
    sample 256 tokens (Taming cin_transformer, greenlist watermark delta=2 gamma=.25 h=1,
    T=1, top-k 250, top-p .92)  ->  codes_to_images  ->  images_to_codes  ->  detect
(and, with more than one rank, the all_gather of codes / counts / p-values: SURVEY 8e's one exchange step).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Weights are seeded random-init tensors of the real architecture (no checkpoints offline);
inputs are resident in HBM when the timed region starts.  Rank 0 prints ONE JSON line.

Test hook (tests/test_bench_distributed_cpu.py): WMAR_BENCH_BACKEND=gloo + WMAR_BENCH_ENGINE="module:factory" run the SAME
control flow (process group, key broadcast, barriers, max-over-ranks timing, gather, JSON line) on CPU with a stand-in engine;
the product engines have no CPU path and the default factory fails loudly without an MI355X.
"""
import argparse
import importlib
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B = 64
GEN = dict(batch_size=B, temperature=1.0, top_k=250, top_p=0.92)
PEAK_F32_MFMA_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TF = 2500.0     # dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E spec
METRIC = "watermarked images/sec at 256x256 (16x16 tok), batch 64; detector p-value delta vs ref"


def usable_cores() -> int:
    """Host threads this process may really use: scheduler affinity capped by the cgroup CPU
    quota, and by 32 (the oracle's GEMMs stop scaling long before that; 256 threads on
    the GPU box's host measured 200x SLOWER than 8)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, min(n, 32))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(gpt_state_gpu, vq_state_gpu, gcfg, vcfg, wm, log, budget_s=75.0, model=None):
    """the CPU oracle (a port of the reference's path, parity-pinned in tests/) on this host's cores,
    fp32, on the two stated configurations -- Taming batch 1 (configs[0]) and batch 64 (configs[1]) -- with the stages timed
    separately: `sample` = 16 decode steps of the full 48-layer model incl. watermark + top-k/top-p + multinomial, extrapolated
    x16 to 256 steps (stated in `sample`); codes_to_images / images_to_codes on 2 images; detect on 64.  3 repeats after one
    warm-up, median and spread reported; repeats are cut (and said so) if the leg would exceed its time budget."""
    from oracle import model_oracle as M
    from oracle import wm_oracle as W

    cores = usable_cores()
    torch.set_num_threads(cores)
    gs = {k: v.cpu() for k, v in gpt_state_gpu.items()}
    vs = {k: v.cpu() for k, v in vq_state_gpu.items()}
    key = W.KeyParams(wm._alive_host, wm._dead_host, gcfg.vocab_size, wm.gamma)
    S = vcfg.codes_size ** 2
    NSTEP = 16
    t_begin = time.perf_counter()

    def timed(fn, reps):
        out = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            out.append(time.perf_counter() - t0)
        return out

    checks = {}

    def sample_cfg(Bc):
        cond = torch.tensor([(i * 37) % 1000 for i in range(Bc)]).view(-1, 1)
        torch.manual_seed(1)
        # the noise of the loop's multinomial draws, drawn up front (one [B, V] exponential per step, the reference's order) so that the
        # SAME run can be replayed on the engine the headline was timed on: the oracle here is the checker of that 48-layer engine
        qs = torch.empty(NSTEP, Bc, gcfg.vocab_size).exponential_(1)
        last = {}
        def run(n):
            last["tok"] = M.sample_with_past(gs, gcfg.n_head, cond, n, 1.0, 250, 0.92, key, 2.0, q_source=lambda i, b, v: qs[i])
        t0 = time.perf_counter()
        run(1)                                                  # warm-up (thread pools, oracle .so)
        warm = time.perf_counter() - t0
        left = budget_s * (0.25 if Bc == 1 else 0.55)
        reps = 3 if warm * NSTEP * 3 < left else 1
        nstep = NSTEP if warm * NSTEP * reps < left * 1.5 else max(1, int(left / max(warm, 1e-3)))
        ts = [t / nstep for t in timed(lambda: run(nstep), reps)]
        if model is not None:
            try:
                eng = model.model.transformer
                got = eng.generate(cond.view(-1).to(model.model.device), nstep, qs[:nstep].contiguous().to(model.model.device), 1.0, 250, 0.92,
                                   wm.wm_ctx(), use_graph=False).cpu()
                ref = last["tok"]
                checks[f"batch{Bc}"] = {"rows": Bc, "steps": nstep, "token_mismatches": int((got != ref).sum()),
                                        "rows_equal": int((got == ref).all(1).sum()), "plan": eng.plan_info(Bc).get("path", "matrix-core plan")}
            except Exception as e:
                checks[f"batch{Bc}"] = {"error": repr(e)}
        return {"s_per_step": statistics.median(ts), "min": min(ts), "max": max(ts), "repeats": reps, "steps_timed": nstep}

    b1 = sample_cfg(1)
    b64 = sample_cfg(B)
    nvq = 2
    full = torch.randint(0, gcfg.vocab_size, (nvq, S))
    img = M.codes_to_images(vs, vcfg, full)                    # warm-up
    vq_reps = 3 if (time.perf_counter() - t_begin) < budget_s * 0.8 else 1
    t_dec = statistics.median(timed(lambda: M.codes_to_images(vs, vcfg, full), vq_reps)) / nvq
    t_enc = statistics.median(timed(lambda: M.images_to_codes(vs, vcfg, img), vq_reps)) / nvq
    det_codes = torch.randint(0, gcfg.vocab_size, (B, S)).numpy()
    W.detect(key, det_codes[:2])
    t_det = statistics.median(timed(lambda: W.detect(key, det_codes), 3)) / B
    per_img_64 = b64["s_per_step"] * S / B + t_dec + t_enc + t_det
    per_img_1 = b1["s_per_step"] * S + t_dec + t_enc + t_det
    log(f"cpu_baseline: B=1 {b1['s_per_step']:.3f} s/step, B=64 {b64['s_per_step']:.3f} s/step, vq decode {t_dec:.2f} + encode {t_enc:.2f} s/img, "
        f"detect {t_det * 1e3:.1f} ms/img, {cores} threads, {time.perf_counter() - t_begin:.0f} s")
    r3 = lambda x: round(x, 4)
    return {"value": 1.0 / per_img_64, "unit": "images/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"Taming batch 64 (configs[1]): {b64['steps_timed']} decode steps of the full 48L model incl. watermark + sampling, "
                      f"{b64['repeats']} repeats (median; x{S}/{b64['steps_timed']} extrapolated to {S} steps), VQGAN decode + encode of "
                      f"{nvq} images ({vq_reps} repeats), detect of {B}; batch 1 (configs[0]) timed the same way.  NOT in this port: the "
                      "reference's per-row Python watermark loop (gentime_watermark.py:229-271, ~150 ms per step at batch 64 = ~38 s per "
                      "64 images, SURVEY section 0) -- the port applies the bias with one vectorised call, so it is FASTER than the reference's CPU path",
            "batch64": {"images_per_s": 1.0 / per_img_64, "sample_s_per_step": r3(b64["s_per_step"]),
                        "sample_s_per_step_min_max": [r3(b64["min"]), r3(b64["max"])], "sample_s_per_image": r3(b64["s_per_step"] * S / B)},
            "batch1": {"images_per_s": 1.0 / per_img_1, "sample_s_per_step": r3(b1["s_per_step"]),
                       "sample_s_per_step_min_max": [r3(b1["min"]), r3(b1["max"])], "sample_s_per_image": r3(b1["s_per_step"] * S)},
            "codes_to_images_s_per_image": r3(t_dec), "images_to_codes_s_per_image": r3(t_enc), "detect_s_per_image": round(t_det, 6),
            # the oracle as the checker of the TIMED engine (48 layers, this run's weights): the timed oracle steps replayed on it, same noise
            "timed_engine_vs_oracle_tokens": checks or None}


def parity_block(log):
    """The 'p-value delta vs ref' half of the metric, computed OUTSIDE the timed region on the committed reference fixtures
    (tests/golden/*.npz: outputs of the reference itself): token mismatches of the 256-step watermarked loop at production
    width, detector p-values of those tokens, decoded pixels of the golden VQGAN.  Data files only -- no oracle involved."""
    import numpy as np
    from wmar_amd.models.engine import GPTEngine, VQGANEngine
    from wmar_amd.utils import synth
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy

    gdir = os.path.join(ROOT, "tests", "golden")
    pv = np.load(os.path.join(gdir, "prod_vectors.npz"))
    gcfg = synth.GPTConfig(vocab_size=16384, block_size=256, n_layer=2, n_head=24, n_embd=1536)
    eng = GPTEngine(gcfg, synth.synth_gpt_state(gcfg, seed=9, logit_scale=10.0), max_batch=64)
    ids = []
    for line in open(os.path.join(ROOT, "wmar_amd", "assets", "vqgan_alive_ids.txt")):
        ids.extend(int(t) for t in line.split(","))
    dead = sorted(set(range(16384)) - set(ids))
    wm = GentimeWatermark({"alive_ids": torch.tensor(ids), "dead_ids": torch.tensor(dead), "embedding": None}, 16384,
                          SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
    torch.manual_seed(11)
    q = torch.stack([torch.empty(4, 16384).exponential_(1) for _ in range(256)]).cuda()
    toks = eng.generate(torch.from_numpy(pv["loop_cond"]).view(-1).cuda(), 256, q, 1.0, 250, 0.92, wm.wm_ctx())
    ref = torch.from_numpy(pv["loop_tokens"].astype(np.int64))
    mism = int((toks.cpu() != ref).sum())
    p = wm.detect(ref.cuda()).cpu().numpy()                    # detector on the reference's own tokens
    dlog = float(np.abs(np.log10(p) - np.log10(pv["loop_pvals"])).max())
    dabs = float(np.abs(p - pv["loop_pvals"]).max())
    seq = torch.from_numpy(pv["gpt_seq"].astype(np.int64)).cuda()
    dlogit = 0.0
    for t in range(2):
        lg = eng.decode_step(seq[:, t], t)
        dlogit = max(dlogit, float(np.abs(lg[:, ::64].cpu().numpy() - pv["gpt_logits"][t]).max()))
    # the VQGAN at the shape the timed step runs (256 x 256, 5 levels, 16384 x 256 codebook): tests/golden/fullsize_vq_vectors.npz holds
    # the reference's own Decoder / Encoder / VectorQuantizer2 outputs on seeded weights (every 4th pixel, 64 pre-quantisation
    # vectors, the re-encoded codes with the reference's best-vs-second margin, the detector's p-values)
    fv = np.load(os.path.join(gdir, "fullsize_vq_vectors.npz"))
    vcfg = synth.TAMING_VQ
    vq = VQGANEngine(vcfg, synth.synth_vq_state(vcfg, seed=31), max_batch=2)
    img = vq.decode(torch.from_numpy(fv["tam_codes"]).cuda())
    sub = np.stack([img[b, :, (b % 4)::4, ((2 * b + 1) % 4)::4].cpu().numpy() for b in range(img.shape[0])])
    dpix = float(np.abs(sub - fv["tam_pixels"]).max())
    codes2_t, pre = vq.encode(img, return_prequant=True)     # the engine's own image (the reference's differs by dpix)
    codes2 = codes2_t.cpu().numpy()
    dpre = float(np.abs(pre.cpu().numpy()[::8] - fv["tam_prequant"]).max())
    bad = np.nonzero(codes2.reshape(-1) != fv["tam_codes_roundtrip"].reshape(-1))[0]
    p2 = wm.detect(codes2_t).cpu().numpy()
    out = {"reference_fixture": "tests/golden/prod_vectors.npz + fullsize_vq_vectors.npz (outputs of the reference's own code)",
           "token_mismatches": mism, "tokens_compared": int(ref.numel()), "max_abs_dlog10_pvalue": dlog, "max_abs_dpvalue": dabs,
           "max_abs_dlogit": dlogit,
           "vqgan_shape": "Taming f16/16384 at 256x256 (the timed shape)", "max_abs_dpixel": dpix, "max_abs_dprequant": dpre,
           "reencoded_code_mismatches": int(len(bad)), "codes_compared": int(codes2.size),
           "mismatches_outside_reference_near_ties": int((fv["tam_margin"][bad] >= 5e-2).sum()),
           "max_abs_dpvalue_after_roundtrip": float(np.abs(p2 - fv["tam_pvals_roundtrip"]).max()) if len(bad) == 0 else None}
    log(f"parity: {out}")
    return out


def small_batch_block(model, wm, gcfg, log):
    """The reference's OWN batch sizes on the same 48-layer engine: BASELINE.json configs[0] (batch 1) and the published run's batch 5
    (configs/taming_generate.json).  Full unit of work -- sample 256 tokens (watermark + top-k + top-p inside the captured loop) ->
    codes_to_images -> images_to_codes -> detect -- one warm-up + best of two.  These rows run the weight-streaming plan of
    wmar_amd/csrc/decode_small.h; the floor is the step's byte stream: 5.54 GB of fp32 weights + the K/V rows of the batch, at 8 TB/s."""
    out = {}
    eng = model.model.transformer
    L, D, V = gcfg.n_layer, gcfg.n_embd, gcfg.vocab_size
    classes = [1, 9, 232, 340, 568, 656, 703, 814, 937, 975]      # configs/taming_generate.json
    for name, Bs in (("taming_b1", 1), ("taming_b5", 5), ("taming_b10", 10)):      # configs[0]; the published run; generate.py's default
        cond = torch.tensor(classes[:Bs], device=model.model.device)
        best = None
        for rep in range(3):
            q = model.draw_noise(256, Bs)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            codes = model.sample(cond, dict(batch_size=Bs, temperature=1.0, top_k=250, top_p=0.92), apply_watermark=True, q=q)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            img = model.codes_to_images(codes); torch.cuda.synchronize(); t2 = time.perf_counter()
            c2 = model.images_to_codes(img); torch.cuda.synchronize(); t3 = time.perf_counter()
            pv = wm.detect(c2); torch.cuda.synchronize(); t4 = time.perf_counter()
            r = {"images_per_s": round(Bs / (t4 - t0), 3), "ms_per_step": round((t1 - t0) / 256 * 1e3, 3), "sample_s": round(t1 - t0, 4),
                 "vq_decode_s": round(t2 - t1, 4), "vq_encode_s": round(t3 - t2, 4), "detect_s": round(t4 - t3, 5)}
            if rep > 0 and (best is None or r["ms_per_step"] < best["ms_per_step"]):
                best = r
        wbytes = 4.0 * (12 * D * D * L + D * V)
        floor_ms = (wbytes + 2.0 * L * D * 4 * 128.5 * Bs) / (PEAK_HBM_GBS * 1e9) * 1e3
        plan = eng.plan_info(Bs)
        roles = {}
        for k, wb in (("qkv", 3 * D * D * 4.0), ("attn", 2.0 * Bs * D * 4 * 128), ("proj", D * D * 4.0), ("fc1", 4 * D * D * 4.0), ("fc2", 4 * D * D * 4.0), ("head", D * V * 4.0)):
            us = eng.profile_role(k, Bs, kv_len=128, iters=2 * L)
            roles[k] = {"kernel": plan.get(k, ""), "avg_us": round(us, 2), "bytes_per_launch": wb, "GBs": round(wb / us * 1e-3, 1),
                        "frac_of_hbm": round(wb / us * 1e-3 / PEAK_HBM_GBS, 3)}
        best.update(config=f"Taming cin_transformer 48L x 1536d + VQGAN f16/16384, 256x256, batch {Bs}, greenlist delta=2 gamma=.25 h=1, "
                           "T=1 top-k 250 top-p 0.92; sample -> decode -> re-encode -> detect", plan=plan.get("path", "matrix-core plan"),
                    step_floor_ms=round(floor_ms, 3), frac_of_step_floor=round(floor_ms / best["ms_per_step"], 3), roles=roles)
        out[name] = best
        log(f"small batch {name}: {best['ms_per_step']} ms per step = {best['frac_of_step_floor']} of the byte-stream floor, {best['images_per_s']} images/s")
    return out


def secondary_block(log, device="cuda"):
    """BASELINE.json configs[2] and configs[3] on the same GPU, AFTER the timed region and after the Taming engines are released
    (driver-verifiable numbers for the two other model families; skip with --no-secondary):
      * RAR-XL 256x256, batch 64 under guidance (128 rows), greenlist watermark and Gumbel key: sample 256 tokens -> MaskGIT-VQGAN
        decode -> re-encode -> detect; floor = max(HBM, fp32 MFMA) per step of SURVEY Appendix F (1.15 ms HBM with every GEMM on the bf16 pipe);
      * Chameleon/Anole-7B text -> image, batch 16 (48 sequences), 1024 image tokens at 512x512, greenlist (FIXED) watermark:
        floor = (13.5 GB of bf16 weights + the bf16 KV rows of the step) / 8 TB/s.
    Random-init weights of the real architectures; one warm-up + one timed repetition each."""
    import gc
    from wmar_amd.utils import synth
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    out = {}
    sync = torch.cuda.synchronize

    def run(model, wm, cond, gen, draw, n_steps):
        best = None
        for rep in range(2):
            q = draw()
            sync(); t0 = time.perf_counter()
            codes = model.sample(cond, gen, apply_watermark=True, q=q) if q is not None else model.sample(cond, gen, apply_watermark=True)
            sync(); t1 = time.perf_counter()
            img = model.codes_to_images(codes); sync(); t2 = time.perf_counter()
            c2 = model.images_to_codes(img); sync(); t3 = time.perf_counter()
            pv = wm.detect(c2); sync(); t4 = time.perf_counter()
            best = {"images_per_s": round(len(cond) / (t4 - t0), 3), "ms_per_step": round((t1 - t0) / n_steps * 1e3, 3),
                    "sample_s": round(t1 - t0, 4), "vq_decode_s": round(t2 - t1, 4), "vq_encode_s": round(t3 - t2, 4), "detect_s": round(t4 - t3, 5),
                    "token_match_after_roundtrip": round(float((c2 == codes).float().mean()), 4), "median_pvalue": float(pv.median())}
        return best

    try:
        from wmar_amd.models.rar_wrapper import RarARMMWrapper
        from wmar_amd.watermarking.gumbel_watermark import GumbelWatermark
        m = RarARMMWrapper.synthetic(max_batch=B)
        cond = torch.arange(B) % 1000
        torch.manual_seed(0)
        wm = GentimeWatermark(m.get_vq(), 1024, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device=device)
        m.set_watermarker(wm)
        r = run(m, wm, cond, None, lambda: m.draw_noise(B), 257)
        # per step (128 rows, cache length 129.5 on average): 3.79 GB of fp32 weights + 5.43 GB of KV at 8 TB/s
        floor_ms = (3.793e9 + 2 * 32 * 1280 * 4 * 129.5 * 128) / (PEAK_HBM_GBS * 1e9) * 1e3
        r.update(config="RAR-XL 256x256, batch 64 x 2 guidance rows, greenlist delta=2 gamma=.25 h=1 + MaskGIT-VQGAN decode / encode + detect",
                 step_floor_ms=round(floor_ms, 3), frac_of_step_floor=round(floor_ms / r["ms_per_step"], 3))
        out["rar_xl_greenlist"] = r
        gw = GumbelWatermark(1024, seed=1234, temperature=1.0, device=device)
        m.set_watermarker(gw)
        r = run(m, gw, cond, None, lambda: None, 257)
        r.update(config="RAR-XL 256x256, batch 64 x 2 guidance rows, Gumbel key (extension: wmar_audio/watermark/engine.py semantics) + VQ decode / encode + detect",
                 step_floor_ms=round(floor_ms, 3), frac_of_step_floor=round(floor_ms / r["ms_per_step"], 3))
        out["rar_xl_gumbel"] = r
        log(f"secondary RAR-XL: {out}")
        del m, wm, gw
        gc.collect(); torch.cuda.empty_cache()
        # the reference's own batch size (configs/rar_generate.json: 10 = 20 rows under guidance): the engine's one-row-tile plan
        Br = 10
        m = RarARMMWrapper.synthetic(max_batch=Br)
        wm = GentimeWatermark(m.get_vq(), 1024, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device=device)
        m.set_watermarker(wm)
        cond_r = torch.arange(Br) % 1000
        bb = None
        for rep in range(2):
            qn = m.draw_noise(Br)
            sync(); t0 = time.perf_counter()
            m.sample(cond_r, None, apply_watermark=True, q=qn)
            sync(); bb = time.perf_counter() - t0
        floor10 = (3.793e9 + 2 * 32 * 1280 * 4 * 129.5 * 2 * Br) / (PEAK_HBM_GBS * 1e9) * 1e3
        out["rar_xl_b10"] = {"ms_per_step": round(bb / 257 * 1e3, 3), "sample_images_per_s": round(Br / bb, 2), "step_floor_ms": round(floor10, 3),
                             "frac_of_step_floor": round(floor10 / (bb / 257 * 1e3), 3),
                             "config": "RAR-XL at the reference's batch size (configs/rar_generate.json: 10 x 2 guidance rows), greenlist; sampling loop only"}
        del m, wm
        gc.collect(); torch.cuda.empty_cache()
    except Exception as e:      # a secondary config must never cost the headline line
        out["rar_xl_error"] = repr(e)
    try:
        from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper
        Bc, n_tok = 16, 1024
        m = ChameleonARMMWrapper.synthetic(vq_cfg=synth.CHAMELEON_VQ, max_batch=Bc)
        wm = GentimeWatermark(m.get_vq(), 65536, SeedStrategy.FIXED, SplitStrategy.RANDOM_STRATIFIED, 0, 2.0, 0.25, device=device)
        m.set_watermarker(wm)
        m.n_image_tokens = n_tok
        text = m.vocab.text_tokens
        cond = [(i, [text[(i * 37 + j * 11) % len(text)] for j in range(12 + i % 5)]) for i in range(Bc)]
        r = run(m, wm, cond, {"temperature": 0.7, "top_p": 0.9}, lambda: m.draw_noise(Bc), n_tok + 17)
        # per step: 13.5 GB of bf16 weights + 48 sequences x 2 x 32 x 4096 x 2 B x (17 + 512) cached rows on average
        floor_ms = (13.48e9 + 48 * 2 * 32 * 4096 * 2 * 529.0) / (PEAK_HBM_GBS * 1e9) * 1e3
        r.update(config="Chameleon/Anole-7B text->image 512x512 (1024 tokens), batch 16 x 3 guidance streams, greenlist (fixed key) "
                        "delta=2 gamma=.25 + VQGAN-512 decode / encode + detect; ms_per_step includes ~17 prompt positions",
                 step_floor_ms=round(floor_ms, 3), frac_of_step_floor=round(floor_ms / r["ms_per_step"], 3))
        out["chameleon_7b"] = r
        log(f"secondary Chameleon-7B: {r}")
        del m, wm
        gc.collect(); torch.cuda.empty_cache()
    except Exception as e:
        out["chameleon_7b_error"] = repr(e)
    return out


def default_engine(device, rank, args):
    """The product path: HIP engines on an MI355X (fails loudly anywhere else)."""
    from wmar_amd.models.taming_wrapper import TamingARMMWrapper
    from wmar_amd.utils import synth
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy

    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: the wmar_amd engines have no CPU implementation")
    gcfg, vcfg = synth.TAMING_GPT, synth.TAMING_VQ
    gs = synth.synth_gpt_state_fast(gcfg, 0, device, logit_scale=30.0)
    vs = synth.synth_vq_state_fast(vcfg, 0, device)
    model = TamingARMMWrapper(None, gpt_cfg=gcfg, vq_cfg=vcfg, gpt_state=gs, vq_state=vs, device=device, max_batch=B)
    model.use_graph = not args.no_graph
    wm = GentimeWatermark(model.get_vq(), gcfg.vocab_size, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1,
                          2.0, 0.25, device=device)
    return model, wm, dict(gs=gs, vs=vs, gcfg=gcfg, vcfg=vcfg)


def gpu_analysis(model, wm, extra, cond, args, world, log):
    """Rank 0, after the timed region: per-role kernel timings (HIP events on the launch stream), rooflines, stage split,
    parity on the golden fixtures, CPU baseline."""
    gcfg, vcfg = extra["gcfg"], extra["vcfg"]
    eng = model.model.transformer
    S = vcfg.codes_size ** 2
    D, V, L = gcfg.n_embd, gcfg.vocab_size, gcfg.n_layer
    # ---- every decode-step kernel role replayed back to back (cycling through the 48 layers, so weights and KV stream from
    # HBM as in a real step) between two hipEvents on the launch stream
    per_step = {"qkv": L, "attn": L, "proj": L, "resid": L + 1, "fc1": L, "fc2": L, "head": 1}
    kv_avg = (S + 1) // 2
    avg_us = {k: eng.profile_role(k, B, kv_len=kv_avg, iters=2 * L) for k in per_step}
    flops = {"qkv": 2.0 * B * 3 * D * D, "proj": 2.0 * B * D * D, "fc1": 2.0 * B * 4 * D * D,
             "fc2": 2.0 * B * 4 * D * D, "head": 2.0 * B * D * V}
    wbytes = {k: flops[k] / (2.0 * B) * 4.0 for k in flops}          # fp32 weights streamed once per launch
    attn_bytes = 2.0 * B * D * 4 * kv_avg                  # K and V rows of kv_avg cached tokens for every (sequence, head)
    share = {k: avg_us[k] * per_step[k] for k in per_step}
    tot = sum(share.values())
    names = eng.plan_info(B)                               # the kernels the engine selects for this batch (wmar_gpt_plan_info)
    per_step["resid"] = int(names.get("resid_launches_per_step", L + 1))       # 1 when the projection launch folds the residual itself
    share = {k: avg_us[k] * per_step[k] for k in per_step}
    tot = sum(share.values())
    roles = {}
    for k in per_step:
        r = {"kernel": names.get(k, k), "avg_us": round(avg_us[k], 2), "launches_per_step": per_step[k],
             "share_of_decode_step": round(share[k] / tot, 3)}
        if k in flops:
            # two ceilings per GEMM role: its weight stream at the HBM spec, and its contraction on the pipe it runs on (the bf16
            # pipe computes an fp32 product as six bf16 piece products: dense bf16 peak / 6 in fp32-equivalent FLOP/s)
            on_bf16 = "bf16 pipe" in r["kernel"]
            pipe_peak = PEAK_BF16_MFMA_TF / 6 if on_bf16 else PEAK_F32_MFMA_TF
            t_hbm = wbytes[k] / (PEAK_HBM_GBS * 1e3)               # us
            t_mfma = flops[k] / (pipe_peak * 1e6)                  # us
            hbm_bound = t_hbm >= t_mfma
            r.update(bound="hbm" if hbm_bound else "mfma",
                     achieved=round(wbytes[k] / avg_us[k] * 1e-3, 1) if hbm_bound else round(flops[k] / avg_us[k] * 1e-6, 2),
                     peak=PEAK_HBM_GBS if hbm_bound else round(pipe_peak, 1), unit="GB/s" if hbm_bound else "TFLOP/s",
                     weight_bytes_per_launch=wbytes[k], flop_per_launch=flops[k],
                     floor_us={"hbm": round(t_hbm, 2), "mfma": round(t_mfma, 2), "pipe": "bf16 x6" if on_bf16 else "fp32"},
                     weight_stream_GBs=round(wbytes[k] / avg_us[k] * 1e-3, 1), frac_of_hbm=round(wbytes[k] / avg_us[k] * 1e-3 / PEAK_HBM_GBS, 4),
                     fp32_equiv_TFLOPs=round(flops[k] / avg_us[k] * 1e-6, 2))
            r["frac"] = round(max(t_hbm, t_mfma) / avg_us[k], 4)
        elif k == "attn":
            r.update(bound="hbm", achieved=round(attn_bytes / avg_us[k] * 1e-3, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                     bytes_per_launch=attn_bytes, kv_len=kv_avg)
            r["frac"] = round(r["achieved"] / PEAK_HBM_GBS, 4)
        roles[k] = r
    dom = max((k for k in roles if "frac" in roles[k]), key=lambda k: share[k])    # ALL roles, attention included
    traffic = None
    try:   # HBM bytes per launch from the committed PMC passes (separate FETCH_SIZE / WRITE_SIZE runs, gfx950 x2 read correction)
        pmc = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{dom}.json")))
        traffic = pmc["hbm_bytes_per_launch"]
    except Exception:
        pass
    roofline = {"bound": roles[dom]["bound"], "achieved": roles[dom]["achieved"], "peak": roles[dom]["peak"],
                "unit": roles[dom]["unit"], "frac": roles[dom]["frac"], "traffic": traffic,
                "traffic_source": f"profiles/pmc_{dom}.json (committed rocprofv3 --pmc passes of this role: FETCH_SIZE x2 + WRITE_SIZE per launch; "
                                  "not re-measured in this run)" if traffic is not None else None,
                "kernel": roles[dom]["kernel"],
                "avg_us": roles[dom]["avg_us"], "launches_per_step": per_step[dom],
                "algorithmic_per_launch": roles[dom].get("bytes_per_launch", roles[dom].get("weight_bytes_per_launch")),
                "share_of_decode_step": roles[dom]["share_of_decode_step"], "role": dom}
    gemm_tf = sum(flops[k] * per_step[k] for k in flops) / sum(share[k] for k in flops) * 1e-6
    torch.cuda.synchronize()
    t1 = time.perf_counter(); codes_t = model.sample(cond, GEN, True); torch.cuda.synchronize()
    t2 = time.perf_counter(); im = model.codes_to_images(codes_t); torch.cuda.synchronize()
    t3 = time.perf_counter(); c2 = model.images_to_codes(im); torch.cuda.synchronize()
    t4 = time.perf_counter(); wm.detect_counts(c2); torch.cuda.synchronize()
    t5 = time.perf_counter()
    split = {"sample_s": round(t2 - t1, 4), "vq_decode_s": round(t3 - t2, 4), "vq_encode_s": round(t4 - t3, 4),
             "detect_s": round(t5 - t4, 5)}
    step_ms = split["sample_s"] / S * 1e3
    # per step max(1.13 ms fp32 MFMA, 1.30 ms HBM) at batch 64; per image 252.7 (decode) / 138.4 + 2.1 (encode) GFLOP
    step_roofline_ms = max(2.0 * B * 1.38412e9 / (PEAK_F32_MFMA_TF * 1e12), (5.53648e9 + 2.0 * L * D * 4 * kv_avg * B) / (PEAK_HBM_GBS * 1e9)) * 1e3
    out = {"roofline": roofline, "roofline_by_role": roles,
           "decode_step": {"ms": round(step_ms, 3), "roofline_ms": round(step_roofline_ms, 3), "frac": round(step_roofline_ms / step_ms, 3),
                           "gemm_TFLOPs_all_roles": round(gemm_tf, 2)},
           # the stride-1 convs with >= 32 input channels run on the bf16 matrix pipe as six bf16 piece products per fp32 product
           # (k_conv_bx): the peak that bounds them is the dense bf16 MFMA peak / 6, in fp32-equivalent TFLOP/s
           "vqgan": {"decode_TFLOPs": round(252.7e9 * B / split["vq_decode_s"] * 1e-12, 1),
                     "encode_TFLOPs": round(140.5e9 * B / split["vq_encode_s"] * 1e-12, 1),
                     "pipe": "v_mfma_f32_32x32x16_bf16, 6 piece products per fp32 product (exact 3-way bf16 split)",
                     "peak": round(PEAK_BF16_MFMA_TF / 6, 1), "peak_fp32_mfma": PEAK_F32_MFMA_TF,
                     "decode_frac": round(252.7e9 * B / split["vq_decode_s"] * 1e-12 / (PEAK_BF16_MFMA_TF / 6), 3),
                     "encode_frac": round(140.5e9 * B / split["vq_encode_s"] * 1e-12 / (PEAK_BF16_MFMA_TF / 6), 3)},
           "end_to_end_roofline": None,
           "stage_seconds_per_batch": split}
    # images/s if every stage ran at its ceiling: 256 decode steps at max(HBM, fp32-MFMA) per step + the VQGAN's 393.2 GFLOP per image
    # (decode 252.7 + encode 138.4 + 2.1 VQ distances) on the bf16 pipe at six piece products per fp32 product
    vq_floor_s = B * (252.7e9 + 140.5e9) / (PEAK_BF16_MFMA_TF / 6 * 1e12)
    roof_s = S * step_roofline_ms * 1e-3 + vq_floor_s
    out["end_to_end_roofline"] = {"images_per_s": round(B / roof_s, 1), "seconds_per_batch": round(roof_s, 4),
                                  "formula": "64 / (256 x max(2 x 64 x 1.384 GFLOP / 157.3 TF, (5.54 GB weights + KV(128.5)) / 8 TB/s) + 64 x 393.2 GFLOP / (2500 / 6 TF))"}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of hipGraph replay")
    ap.add_argument("--no-secondary", action="store_true", help="skip the RAR-XL / Chameleon-7B numbers (after the timed region)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    backend = os.environ.get("WMAR_BENCH_BACKEND", "nccl")          # "nccl" IS RCCL on ROCm
    on_gpu = backend == "nccl"
    if on_gpu:
        torch.cuda.set_device(local_rank)
    device = f"cuda:{local_rank}" if on_gpu else "cpu"
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend)  # RCCL over xGMI

    def log(msg):
        if rank == 0:
            print(msg, file=sys.stderr, flush=True)

    def sync():
        if on_gpu:
            torch.cuda.synchronize()

    factory = default_engine
    if os.environ.get("WMAR_BENCH_ENGINE"):
        mod, fn = os.environ["WMAR_BENCH_ENGINE"].split(":")
        factory = getattr(importlib.import_module(mod), fn)
    t0 = time.time()
    model, wm, extra = factory(device, rank, args)
    # the key table is built once (host MT19937 + Fisher-Yates over all context sums) on rank 0 and broadcast over RCCL
    if world > 1:
        from wmar_amd import harness
        harness.broadcast_key_table(wm, device)
    else:
        wm.key_table()
    model.set_watermarker(wm)
    if on_gpu:
        _ = model.model.vq_engine
        sync()
        log(f"setup {time.time() - t0:.1f}s; GPT engine {model.model.transformer.device_bytes / 1e9:.1f} GB, "
            f"VQGAN engine {model.model.vq_engine.device_bytes / 1e9:.1f} GB")

    # weak scaling: every rank generates its own batch of 64 class labels (rank = the reference's chunk id, generate.py:204,304)
    cond = torch.tensor([((rank * B + i) * 37) % 1000 for i in range(B)], device=device)
    torch.manual_seed(1 + 1000 * rank)
    if on_gpu:
        torch.cuda.manual_seed_all(1 + 1000 * rank)

    def gather(t):
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t.contiguous())
        return torch.cat(outs)

    def step():
        codes = model.sample(cond, GEN, apply_watermark=True)
        imgs = model.codes_to_images(codes)
        codes2 = model.images_to_codes(imgs)
        pv, ns, ng = wm.detect_counts(codes2)
        if world > 1:      # the path's one exchange step: codes int64[64,256], counts int32[64], p-values f64[64] per rank
            codes2_all, ns_all, ng_all, pv_all = gather(codes2), gather(ns), gather(ng), gather(pv)
            return (pv_all, ns_all, ng_all), codes, codes2, codes2_all
        return (pv, ns, ng), codes, codes2, codes2

    for _ in range(args.warmup):
        step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        (pv, ns, ng), codes, codes2, codes2_all = step()
    sync()
    if world > 1:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    value = world * B * args.steps / dt

    if rank == 0:
        S = codes.shape[1]
        out = {
            "metric": METRIC, "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "Taming cin_transformer (48L x 1536d x 24h, V=16384) + VQGAN f16/16384, 256x256, "
                                   "batch 64 per GPU, greenlist watermark delta=2 gamma=0.25 h=1 (linear/stratifiedrand), "
                                   "T=1 top-k 250 top-p 0.92; sample -> decode -> re-encode -> detect; random-init weights",
                       "batch_per_gpu": B, "tokens_per_image": S, "parallelism": f"replicas x{world} (images sharded, no data-path "
                       "collective; codes / counts / p-values all-gathered once per step)", "decode_loop": "eager" if args.no_graph else "hipGraph replay",
                       "backend": backend},
            "gathered": {"codes": list(codes2_all.shape), "pvalues": int(pv.numel())},
            "detector": {"n_scored_mean": float(ns.float().mean()), "n_green_mean": float(ng.float().mean()),
                         "token_match_after_roundtrip": float((codes == codes2).float().mean())},
        }
        if on_gpu and factory is default_engine:
            out.update(gpu_analysis(model, wm, extra, cond, args, world, log))
            out["end_to_end_frac_of_roofline"] = round(value / world / out["end_to_end_roofline"]["images_per_s"], 3)
            out["parity"] = None if args.no_parity else parity_block(log)
            out["cpu_baseline"] = None
            if world == 1 and not args.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(extra["gs"], extra["vs"], extra["gcfg"], extra["vcfg"], wm, log, model=model)
            out["secondary"] = None
            small = None
            if world == 1 and not args.no_secondary:
                try:
                    small = small_batch_block(model, wm, extra["gcfg"], log)
                except Exception as e:      # a secondary config must never cost the headline line
                    small = {"taming_small_batch_error": repr(e)}
            if world == 1 and not args.no_secondary:
                # the Taming engines (22.6 GB) are released first: the 7B model + its KV cache need 40 GB of their own
                import gc
                del model, wm, extra
                gc.collect(); torch.cuda.empty_cache()
                out["secondary"] = secondary_block(log, device)
                if small:
                    out["secondary"].update(small)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
