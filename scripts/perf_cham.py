"""Chameleon-7B (BASELINE config 4) timing on one GPU: text->image token loop at batch B (3B sequences),
then VQGAN-512 decode / re-encode / detect.  Synthetic bf16 weights."""
import os, sys, time
sys.path.insert(0, os.environ.get("WMAR_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper
from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
from wmar_amd.utils import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n_tok = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
with_vq = (sys.argv[3] != "0") if len(sys.argv) > 3 else True
t0 = time.perf_counter()
vq_cfg = synth.CHAMELEON_VQ if with_vq else synth.VQConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(), resolution=64,
                                                             z_channels=32, embed_dim=32, n_embed=8192)
m = ChameleonARMMWrapper.synthetic(vq_cfg=vq_cfg, max_batch=B)
torch.cuda.synchronize()
print(f"engine up in {time.perf_counter()-t0:.1f}s, {m.model.engine.device_bytes/2**30:.1f} GiB", flush=True)
wm = GentimeWatermark(m.get_vq(), 65536, SeedStrategy.FIXED, SplitStrategy.RANDOM_STRATIFIED, 0, 2.0, 0.25, device="cuda")
m.set_watermarker(wm)
m.n_image_tokens = n_tok
text = m.vocab.text_tokens
cond = [(i, [text[(i * 37 + j * 11) % len(text)] for j in range(12 + i % 5)]) for i in range(B)]
for rep in range(2):
    q = m.draw_noise(B); torch.cuda.synchronize()
    t0 = time.perf_counter()
    if n_tok != 1024:
        m.is_codes_shaped = lambda c: True
    codes = m.sample(cond, {"temperature": 0.7, "top_p": 0.9}, apply_watermark=True, q=q)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print(f"rep{rep}: sample {t1-t0:.3f}s = {(t1-t0)/(n_tok+17)*1e3:.3f} ms/step (incl. ~17 prefill steps)", flush=True)
    if with_vq and n_tok == 1024:
        img = m.codes_to_images(codes); torch.cuda.synchronize(); t2 = time.perf_counter()
        c2 = m.images_to_codes(img); torch.cuda.synchronize(); t3 = time.perf_counter()
        pv = wm.detect(c2); torch.cuda.synchronize(); t4 = time.perf_counter()
        print(f"      decode {t2-t1:.3f}s encode {t3-t2:.3f}s detect {t4-t3:.4f}s total {t4-t0:.3f}s -> {B/(t4-t0):.2f} img/s  "
              f"match {(c2==codes).float().mean().item():.3f} p_med {pv.median().item():.2e}", flush=True)
