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
    for aug_name, aug_fn, aug_params in eval_params.get("augmentations", []):
        batch_log[key][aug_name] = []
        for aug_param in aug_params:
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
                                        param=param, metrics=metrics))
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


def generate_sharded(outdir, model, all_inputs, watermarker, eval_params, gen_params, seed: int):
    """One process per GPU: rank r == the reference's ``--chunk_id r --num_chunks world_size``
    (generate.py:204 batch striping, :304 seed offset).  No data-path collective; the result
    records are gathered on rank 0."""
    import torch.distributed as dist

    rank = dist.get_rank() if dist.is_initialized() else 0
    world = dist.get_world_size() if dist.is_initialized() else 1
    seed_everything(seed, rank)
    recs = generate(outdir, model, all_inputs, watermarker, eval_params, gen_params, chunk_id=rank, num_chunks=world)
    if world == 1:
        return recs
    gathered: List[Optional[list]] = [None] * world if rank == 0 else None
    dist.gather_object(recs, gathered, dst=0)
    if rank != 0:
        return None
    out = [r for part in gathered for r in part]
    out.sort(key=lambda r: (r["batch_idx"], r["idx"], r["transform"], str(r["param"])))
    return out
