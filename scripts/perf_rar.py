"""RAR-XL end-to-end timing on one GPU: sample (CFG, 256 steps) -> decode -> re-encode -> detect."""
import sys, time, os
sys.path.insert(0, os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wmar_amd.models.rar_wrapper import RarARMMWrapper
from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
m = RarARMMWrapper.synthetic(max_batch=B)
wm = GentimeWatermark(m.get_vq(), 1024, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 2.0, 0.25, device="cuda")
m.set_watermarker(wm)
cond = torch.arange(B) % 1000
torch.manual_seed(0)
def sync(): torch.cuda.synchronize()
for r in range(reps + 1):
    q = m.draw_noise(B); sync()
    t0 = time.perf_counter(); codes = m.sample(cond, None, True, q=q); sync()
    t1 = time.perf_counter(); img = m.codes_to_images(codes); sync()
    t2 = time.perf_counter(); c2 = m.images_to_codes(img); sync()
    t3 = time.perf_counter(); pv = wm.detect(c2); sync()
    t4 = time.perf_counter()
    print(f"rep{r}: sample {t1-t0:.4f}s ({(t1-t0)/256*1e3:.3f} ms/step) decode {t2-t1:.4f}s encode {t3-t2:.4f}s detect {t4-t3:.4f}s "
          f"total {t4-t0:.4f}s -> {B/(t4-t0):.1f} img/s  match {(c2==codes).float().mean().item():.3f} p_med {pv.median().item():.2e}", flush=True)
