"""The generation/evaluation harness of the reference's ``generate.py`` (generate.py:37-232),
kept to its contract: same batching and chunking rule, same ``batch_log`` structure, same
output file layout -- with two changes that matter on MI355X:

* detection of a whole batch is ONE kernel launch (``detect_batch``) instead of one
  ``watermarker.detect`` call per image (wmar/utils/metrics.py:43);
* ``generate_sharded`` maps the reference's ``--chunk_id/--num_chunks`` job array onto the
  ranks of one ``torch.distributed`` job (one process per GPU, RCCL): rank r plays chunk r,
  and the per-image results are gathered on rank 0.
"""
from __future__ import annotations

import json
import os
import random
import time
from typing import List, Optional

import numpy as np
import torch

from .utils.metrics import compute_metric
from .utils.utils import chw_to_pillow


def make_batches(all_inputs, batch_size):
    """generate.py:180-185 -- consecutive batches, the last one may be smaller."""
    batches = []
    for i in range(len(all_inputs) // batch_size):
        batches.append(all_inputs[i * batch_size:(i + 1) * batch_size])
    if len(all_inputs) % batch_size != 0:
        batches.append(all_inputs[(len(all_inputs) // batch_size) * batch_size:])
    return batches


def seed_everything(seed: int, chunk_id: int = 0):
    """generate.py:304-308 -- seed + 1000*chunk_id for random / numpy / torch / torch.cuda."""
    s = seed + (1000 * chunk_id)
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(s)
    return s


@torch.no_grad()
def fill_batch_log(batch_log, key, model, codes, eval_params, sync_manager=None):
    """generate.py:112-164 without the synchronization layer: decode, `max_roundtrips` encode->decode round trips, then
    every (transform, parameter) of the augmentation table on the WHOLE batch (one re-encode launch sequence per parameter)."""
    assert sync_manager is None, "the WAM/SyncSeal layer is outside the MI355X hot path"
    imgs = model.codes_to_images(codes)  # [b, 3, R, R] in [-1, 1]
    batch_log[key] = {}
    batch_log[key]["roundtrips"] = [(0, codes.cpu().numpy(), imgs.cpu().numpy(), None)]
    curr_imgs = imgs
    for T in range(1, eval_params["max_roundtrips"] + 1):
        curr_codes = model.images_to_codes(curr_imgs)
        curr_imgs = model.codes_to_images(curr_codes)
        batch_log[key]["roundtrips"].append((T, curr_codes.cpu().numpy(), curr_imgs.cpu().numpy(), None))
    from .augmentations import device_ops as _dev_aug
    # the fused form is the default table's arithmetic: only for the AugmentationManager's own entries (a caller's own callable under
    # one of its names keeps its callable)
    fused_ok = eval_params.get("fuse_augmentations", True)
    for aug_name, aug_fn, aug_params in eval_params.get("augmentations", []):
        batch_log[key][aug_name] = []
        for aug_param in aug_params:
            # one launch per (transform, parameter) over the whole batch where the transform has a device form (csrc/augment.hip:
            # range change, transform, clamp and range change back fused; bit-identical to the sequence below), else the module
            aug_imgs = _dev_aug.fused(aug_name, imgs, aug_param) if (fused_ok and getattr(aug_fn, "_wmar_default", False)) else None
            if aug_imgs is None:
                imgs_zero_to_one = imgs / 2.0 + 0.5
                aug_imgs = aug_fn(imgs_zero_to_one, aug_param).clamp(0, 1) * 2.0 - 1.0
            aug_codes = model.images_to_codes(aug_imgs)
            batch_log[key][aug_name].append((aug_param, aug_codes.cpu().numpy(), aug_imgs.cpu().numpy(), None))


def compute_metrics_and_save_from_batch_log(log, outdir, watermarker, eval_params, cond_indices, compressors=None):
    """generate.py:37-108: per (method, transform, param, image): metrics json + png + npy."""
    results = []
    for method in log.keys() - ["batch"]:
        orig_codes = log[method]["roundtrips"][0][1]
        orig_imgs = [chw_to_pillow(img) for img in log[method]["roundtrips"][0][2]]
        for transform in log[method].keys():
            for _, (param, codes, imgs, imgs_nosync) in enumerate(log[method][transform]):
                # batched detection: one launch for the whole batch, then per-image bookkeeping
                pvals = None
                if "pvalue" in eval_params["metric_names"] and watermarker is not None:
                    pvals = watermarker.detect(torch.from_numpy(np.ascontiguousarray(codes))).cpu().numpy()
                for i in range(len(codes)):
                    conditioning = log["batch"][i]
                    if isinstance(conditioning, torch.Tensor):
                        conditioning = conditioning.detach().cpu().item()
                    if isinstance(conditioning, tuple):
                        conditioning = conditioning[0]
                    code, orig_code = codes[i], orig_codes[i]
                    img = chw_to_pillow(imgs[i])
                    metrics = {}
                    for metric_name in eval_params["metric_names"]:
                        if metric_name == "pvalue" and pvals is not None:
                            metrics[metric_name] = float(pvals[i])
                        else:
                            metrics[metric_name] = compute_metric(metric_name, code, orig_code, img, orig_imgs[i],
                                                                  watermarker, transform, param, compressors=compressors)
                    cond_index = cond_indices[i]
                    results.append(dict(conditioning=conditioning, idx=cond_index, method=method, transform=transform,
                                        param=param, metrics=metrics, codes=np.asarray(code, dtype=np.int64)))
                    if outdir is None:
                        continue
                    if not eval_params["orig_only"]:
                        curr_outdir = os.path.join(outdir, f"c={conditioning},idx={cond_index}")
                        os.makedirs(curr_outdir, exist_ok=True)
                        stem = os.path.join(curr_outdir, f"{cond_index:04}_{method}_{transform}_{param}")
                        img.save(stem + ".png")
                        np.save(stem + ".npy", code)
                        with open(stem + ".json", "w") as f:
                            json.dump(metrics, f)
                    else:
                        assert param == 0 and transform == "roundtrips"
                        os.makedirs(os.path.join(outdir, "images"), exist_ok=True)
                        os.makedirs(os.path.join(outdir, "codes"), exist_ok=True)
                        suffix = f"_{method}" if len(log.keys()) > 2 else ""
                        img.save(os.path.join(outdir, "images", f"{conditioning}:{cond_index:04}{suffix}.png"))
                        np.save(os.path.join(outdir, "codes", f"{conditioning}:{cond_index:04}{suffix}.npy"), code)
    return results


@torch.no_grad()
def generate(outdir, model, all_inputs, watermarker, eval_params, gen_params, chunk_id=0, num_chunks=1,
             compressors=None, sync_manager=None):
    """generate.py:168-232.  Returns the per-image result records (the reference returns None
    and only writes files; the records make the multi-GPU gather possible)."""
    batches = make_batches(all_inputs, gen_params["batch_size"])
    base_count_per_conditioning = {}
    results = []
    for batch_idx, batch in enumerate(batches):
        cond_indices = []
        for c in batch:
            if isinstance(c, torch.Tensor):
                c = c.detach().cpu().item()
            if isinstance(c, tuple):
                c = c[0]
            base_count_per_conditioning[c] = base_count_per_conditioning.get(c, 0) + 1
            cond_indices.append(base_count_per_conditioning[c])
        if batch_idx % num_chunks != chunk_id:
            continue  # counts are advanced for skipped batches too (generate.py:193-207)
        t_start = time.time()
        codes = model.sample(batch, gen_params, apply_watermark=watermarker is not None)
        all_codes = {str(watermarker): codes}
        if codes.is_cuda:       # generation is enqueued asynchronously (hipGraph replays): wait before reading the clock
            torch.cuda.synchronize(codes.device)
        sample_s = time.time() - t_start
        batch_log = {"batch": batch}
        for key, cd in all_codes.items():
            fill_batch_log(batch_log, key, model, cd, eval_params, sync_manager=sync_manager)
        recs = compute_metrics_and_save_from_batch_log(batch_log, outdir, watermarker, eval_params,
                                                       cond_indices=cond_indices, compressors=compressors)
        for r in recs:
            r["batch_idx"] = batch_idx
            r["sample_seconds"] = sample_s
        results.extend(recs)
    return results


_REC_COLS = 9   # batch_idx, conditioning, idx, combo, pvalue, l0, psnr, sample_seconds, has-metric bits


def _combos(eval_params):
    """(transform, param) pairs in the order fill_batch_log produces them -- identical on every rank."""
    out = [("roundtrips", T) for T in range(0, eval_params["max_roundtrips"] + 1)]
    for aug_name, _, aug_params in eval_params.get("augmentations", []):
        out += [(aug_name, p) for p in aug_params]
    return out


def gather_records(recs, eval_params, device, dst=0):
    """The per-image results of all ranks on rank `dst`, exchanged as TENSORS with ``all_gather`` (RCCL over xGMI when the process
    group is "nccl"; SURVEY 8e: codes int64[n, L], p-values f64[n], plus the bookkeeping columns) -- no pickling on the data path.
    Method names and the (transform, param) table are the same on every rank and are not sent."""
    import torch.distributed as dist

    rank, world = dist.get_rank(), dist.get_world_size()
    unknown = set(eval_params["metric_names"]) - {"pvalue", "l0", "psnr"}
    assert not unknown, f"gather_records carries pvalue / l0 / psnr only, not {sorted(unknown)}"
    combos = _combos(eval_params)
    index = {(t, str(p)): i for i, (t, p) in enumerate(combos)}
    n = len(recs)
    L = max([len(r["codes"]) for r in recs], default=0)
    from . import comm as _comm
    rc = _comm.active(device)           # WMAR_COMM=rccl: the C-ABI RCCL wrappers instead of torch.distributed (same exchange)

    def _all_gather(t):
        if rc is not None:
            return rc.all_gather(t)
        outs = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(outs, t)
        return outs

    meta = torch.tensor([n, L], dtype=torch.int64, device=device)
    metas = _all_gather(meta)
    n_max = max(int(m[0]) for m in metas)
    L_max = max(int(m[1]) for m in metas)
    rows = torch.zeros(n_max, _REC_COLS, dtype=torch.float64)
    codes = torch.zeros(n_max, max(L_max, 1), dtype=torch.int64)
    for i, r in enumerate(recs):
        m = r["metrics"]
        bits = sum(1 << j for j, k in enumerate(("pvalue", "l0", "psnr")) if m.get(k) is not None)
        cond = r["conditioning"]
        assert isinstance(cond, (int, float)) or hasattr(cond, "__int__"), f"conditioning {cond!r} must be numeric to be gathered (prompt tuples: pass their index)"
        rows[i] = torch.tensor([r["batch_idx"], float(cond), r["idx"], index[(r["transform"], str(r["param"]))],
                                m["pvalue"] if bits & 1 else 0.0, m["l0"] if bits & 2 else 0.0, m["psnr"] if bits & 4 else 0.0,
                                r["sample_seconds"], bits], dtype=torch.float64)
        codes[i, :len(r["codes"])] = torch.from_numpy(r["codes"])
    rows, codes = rows.to(device), codes.to(device)
    all_rows = _all_gather(rows)
    all_codes = _all_gather(codes)
    if rank != dst:
        return None
    method = recs[0]["method"] if recs else None
    out = []
    for w in range(world):
        nw, Lw = int(metas[w][0]), int(metas[w][1])
        rw, cw = all_rows[w].cpu(), all_codes[w].cpu()
        for i in range(nw):
            bits = int(rw[i, 8])
            t, p = combos[int(rw[i, 3])]
            metrics = {k: (float(rw[i, 4 + j]) if bits & (1 << j) else None) for j, k in enumerate(("pvalue", "l0", "psnr"))}
            metrics = {k: v for k, v in metrics.items() if k in eval_params["metric_names"]}
            out.append(dict(conditioning=int(rw[i, 1]), idx=int(rw[i, 2]), method=method, transform=t, param=p, metrics=metrics,
                            codes=cw[i, :Lw].numpy().copy(), batch_idx=int(rw[i, 0]), sample_seconds=float(rw[i, 7])))
    if method is None:
        for r in out:
            r.pop("method")
    out.sort(key=lambda r: (r["batch_idx"], r["idx"], r["transform"], str(r["param"])))
    return out


def generate_sharded(outdir, model, all_inputs, watermarker, eval_params, gen_params, seed: int):
    """One process per GPU: rank r == the reference's ``--chunk_id r --num_chunks world_size``
    (generate.py:204 batch striping, :304 seed offset).  No data-path collective; the result
    records (codes, p-values, l0, psnr) are all-gathered as tensors and assembled on rank 0."""
    import torch.distributed as dist

    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    seed_everything(seed, rank)
    recs = generate(outdir, model, all_inputs, watermarker, eval_params, gen_params, chunk_id=rank, num_chunks=world)
    if world == 1:        # the same record shape as the gathered one: only the requested metrics
        for r in recs:
            r["metrics"] = {k: v for k, v in r["metrics"].items() if k in eval_params["metric_names"]}
        return recs
    for r in recs:          # the method string is rank-independent; conditioning of a prompt tuple is its index
        if isinstance(r["conditioning"], tuple):
            r["conditioning"] = r["conditioning"][0]
    dev = model.device if dist.get_backend() == "nccl" else torch.device("cpu")
    return gather_records(recs, eval_params, dev)


def broadcast_key_table(watermarker, device, src=0):
    """Build the key table once (rank `src`: MT19937 + Fisher-Yates over all context sums on the host cores) and broadcast the
    finished bitmap (RCCL broadcast on "nccl"): 32 MiB for Taming h=1 instead of `world` rebuilds."""
    import torch.distributed as dist

    if dist.get_rank() == src:
        # the builder is host code either way; on a CPU process group (gloo: the N > 1 tests) the finished bitmap stays on the host
        on_gpu = torch.device(device).type == "cuda"
        host_build = not on_gpu and hasattr(watermarker, "key_table_host")
        table = torch.from_numpy(watermarker.key_table_host().view(np.int32).copy()) if host_build else watermarker.key_table()
        shape = torch.tensor(list(table.shape), dtype=torch.int64, device=device)
    else:
        shape = torch.zeros(2, dtype=torch.int64, device=device)
    from . import comm as _comm
    rc = _comm.active(device)
    bcast = (lambda t: rc.broadcast_(t, src)) if rc is not None else (lambda t: dist.broadcast(t, src))
    bcast(shape)
    if dist.get_rank() != src:
        table = torch.empty(tuple(shape.tolist()), dtype=torch.int32, device=device)
    bcast(table)
    watermarker.set_key_table(table)
    return table
