"""f4: Chameleon's interleaved text + image mode (wmar/models/chameleon_wrapper.py:47-134 -> chameleon.py:214-297, 392-440).
* split_token_sequence against hand cases (pure host logic);
* the text-mode logits chain (text watermark -> allow-only -> <boi> cut-off -> HF repetition penalty -> temperature -> top-p ->
  multinomial) against outputs of the reference's own processor classes (tests/golden/chameleon_vectors.npz cham_text_*);
* sample_interleaved end to end on a small model: segment structure, determinism, captured == eager image phases."""
import os

import numpy as np
import pytest
import torch

from tests.conftest import REPO
from wmar_amd.utils import synth


def test_split_token_sequence_hand_cases():
    from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper as W
    boi, eoi = 100, 101
    t = lambda *a: torch.tensor([list(a)])
    segs = W.split_token_sequence(t(1, 2, boi, 7, 8, 9, eoi, 3, boi, 5), boi, eoi)
    assert [(k, v.tolist()) for k, v in segs] == [("text_seg", [[1, 2]]), ("image_seg", [[7, 8, 9]]), ("text_seg", [[3]]), ("image_seg", [[5]])]
    assert [(k, v.tolist()) for k, v in W.split_token_sequence(t(boi, 4, eoi), boi, eoi)] == [("image_seg", [[4]])]
    # an <eoi> outside an image is ordinary text; an empty image leaves an empty segment; trailing text is kept
    assert [(k, v.tolist()) for k, v in W.split_token_sequence(t(eoi, 1, boi, eoi, 2), boi, eoi)] == \
        [("text_seg", [[eoi, 1]]), ("image_seg", [[]]), ("text_seg", [[2]])]
    with pytest.raises(AssertionError):
        W.split_token_sequence(torch.zeros(2, 3, dtype=torch.long), boi, eoi)


def _small(max_prompt_len=16, seed_state=8):
    from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper
    cfg = synth.ChameleonConfig(dim=256, n_layers=2, n_heads=4, n_kv_heads=4, vocab_size=2048, multiple_of=64)
    vq_cfg = synth.VQConfig(ch=32, ch_mult=(1, 2), num_res_blocks=1, attn_resolutions=(), resolution=16, z_channels=32, embed_dim=32,
                            n_embed=512)
    sd = synth.synth_chameleon_state(cfg, seed=seed_state, logit_scale=6.0)
    vm = synth.synth_chameleon_vocab(2048, 512)
    return ChameleonARMMWrapper(None, 0, cfg=cfg, state=sd, vocab_map=vm, vq_cfg=vq_cfg, vq_state=synth.synth_vq_state(vq_cfg, 1),
                                max_batch=2, max_prompt_len=max_prompt_len)


@pytest.mark.gpu
@pytest.mark.parametrize("name,seed,h", [("wm", "linear", 1), ("nowm_limit", None, 0), ("fixed", "fixed", 0)])
def test_text_chain_equals_reference_processors(name, seed, h):
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    cv = np.load(os.path.join(REPO, "tests", "golden", "chameleon_vectors.npz"))
    m = _small()
    if seed is not None:
        m.set_watermarker(None, GentimeWatermark(m.get_vq(), 2048, SeedStrategy(seed), SplitStrategy.RANDOM_STRATIFIED, h, 2.0, 0.25, device="cuda"))
    assert m._allow_text.cpu().tolist() == cv["cham_text_allowed"].tolist()
    lg = torch.from_numpy(cv["cham_text_logits"]).cuda()
    ids = torch.from_numpy(cv["cham_text_input_ids"]).cuda()
    tok = m.text_logits_chain(ids, lg, torch.from_numpy(cv["cham_text_q"]).cuda(), temperature=0.7, top_p=0.9, repetition_penalty=1.2,
                              boi_limit=int(cv[f"cham_text_{name}_limit"]), apply_watermark=seed is not None)
    assert tok.cpu().tolist() == cv[f"cham_text_{name}_tok"].tolist()
    # the in-place part of the chain (before temperature / top-p) agrees with the reference where the reference kept the entry
    ref = cv[f"cham_text_{name}_processed"]
    kept = np.isfinite(ref)
    np.testing.assert_allclose((lg.cpu().numpy() / np.float32(0.7))[kept], ref[kept], rtol=1e-6, atol=1e-6)


@pytest.mark.gpu
def test_sample_interleaved_end_to_end():
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy
    m = _small(max_prompt_len=200)          # max_seq_len = 200 + 64: room for two images
    v = m.vocab
    m._allow_text = torch.tensor([v.eos_id, v.begin_image], dtype=torch.int64, device="cuda")   # every text step: <boi> or <eos>
    wm = GentimeWatermark(m.get_vq(), 2048, SeedStrategy.LINEAR, SplitStrategy.RANDOM_STRATIFIED, 1, 3.0, 0.25, device="cuda")
    m.set_watermarker(wm, None)
    text = v.text_tokens
    gp = {"temperature": 0.9, "top_p": 0.8}
    # prompts after which this random model prefers <boi> to <eos> (so that the text decoder hands over to the image decoder)
    margins = []
    for c in range(40):
        p = [text[(7 * c + 3) % len(text)], text[(13 * c + 50) % len(text)]]
        row = m.tokens_from_ui([{"type": "ids", "value": p}, {"type": "sentinel", "value": "<END-OF-TURN>"}])
        lg = m._prefill([row])[0]
        margins.append((float(lg[v.begin_image] - lg[v.eos_id]), p))
    prompts = [p for _, p in sorted(margins, reverse=True)[:3]]
    assert sorted(margins, reverse=True)[0][0] > 0
    runs = {}
    for graph in (True, False):
        m.use_graph = graph
        for i, p in enumerate(prompts):
            m.seed = i
            segs = m.sample_interleaved([(0, p)], gp, apply_watermark=True, max_gen_len=150)
            runs[(graph, i)] = [(k, t.cpu().tolist()) for k, t in segs]
    for i in range(len(prompts)):
        assert runs[(True, i)] == runs[(False, i)]                            # captured image loop == eager loop inside the mode
    m.use_graph, m.seed = True, 1
    again = m.sample_interleaved([(0, prompts[1])], gp, apply_watermark=True, max_gen_len=150)
    assert [(k, t.cpu().tolist()) for k, t in again] == runs[(True, 1)]         # same seed, same sequence
    n_images = 0
    for (graph, i), segs in runs.items():
        for kind, toks in segs:
            if kind == "image_seg":
                n_images += 1
                assert len(toks[0]) == 64 and set(toks[0]) <= set(v.image_tokens)
                assert wm.detect(torch.tensor(toks).cuda()).shape == (1,)
            else:
                assert set(toks[0]) <= {v.eos_id}
    assert n_images >= 2                                                            # the mode did switch decoders
