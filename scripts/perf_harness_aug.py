"""Dev tool: the harness's evaluation sweep (generate.py:112-164: decode, one round trip, every (transform, parameter) of the default
AugmentationManager table -> images_to_codes) on the full-size Taming tokenizer, images per second; WMAR_AUG_TORCH=1 runs the torch
restatements on the device instead of the kernels of csrc/augment.hip.  usage: perf_harness_aug.py [batch=16] [jpeg=0]"""
import os, sys, time
import torch
ROOT = os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wmar_amd import harness
from wmar_amd.augmentations import AugmentationManager
from wmar_amd.models.engine import VQGANEngine
from wmar_amd.utils import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
with_jpeg = len(sys.argv) > 2 and sys.argv[2] == "1"
cfg = synth.TAMING_VQ
eng = VQGANEngine(cfg, synth.synth_vq_state_fast(cfg, 0, "cuda"), max_batch=B)


class M:                       # the two wrapper calls fill_batch_log makes
    def codes_to_images(self, codes): return eng.decode(codes)
    def images_to_codes(self, imgs): return eng.encode(imgs)


augs = [a for a in AugmentationManager(False, False, True).augs if with_jpeg or a[0] != "jpeg"]
n_pairs = sum(len(p) for _, _, p in augs)
ev = {"metric_names": [], "augmentations": augs, "max_roundtrips": 1, "orig_only": False}
codes = torch.randint(0, cfg.n_embed, (B, 256), device="cuda")
for it in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    log = {}
    harness.fill_batch_log(log, "m", M(), codes, ev)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"batch {B}, {n_pairs} (transform, parameter) pairs{' incl. jpeg' if with_jpeg else ''}, kernels={'torch' if os.environ.get('WMAR_AUG_TORCH') else 'hip'}: "
          f"{dt:.3f} s per batch = {B / dt:.2f} images/s through the whole sweep ({B * (n_pairs + 2) / dt:.0f} encodes/s)")
