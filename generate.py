"""CLI of the reference's ``generate.py`` (generate.py:235-426) for the MI355X build.

Same flags and model dispatch (`--model taming | rar | chameleon7b`, generate.py:319-327 of the reference).
Launch one process per GPU to shard the batches the way the reference's
``--chunk_id/--num_chunks`` job array does:

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 generate.py ...

`--synthetic 1` swaps the (absent) checkpoints for seeded random-init weights.
"""
import argparse
import json
import os
import sys

import torch


def get_parser():
    def str2bool(v):
        if isinstance(v, bool):
            return v
        if v.lower() in ("yes", "true", "t", "y", "1"):
            return True
        elif v.lower() in ("no", "false", "f", "n", "0"):
            return False
        raise argparse.ArgumentTypeError("Boolean value expected.")

    parser = argparse.ArgumentParser()
    parser.add_argument("--outdir", type=str, help="where to save the samples")
    parser.add_argument("--model", type=str, choices=["taming", "chameleon7b", "rar"], help="model to use")
    parser.add_argument("--modelpath", type=str, help="path to the model (see README.md)")
    parser.add_argument("--encoder_ft_ckpt", type=str, help="path to the encoder patch")
    parser.add_argument("--decoder_ft_ckpt", type=str, help="path to the decoder patch")
    parser.add_argument("--num_samples_per_conditioning", type=int, help="samples per imgnet class or coco prompt")
    parser.add_argument("--conditioning", type=str, help="comma-sep classes (imagenet) or coco txt file")
    parser.add_argument("--batch_size", type=int, nargs="?", help="batch size", default=10)
    parser.add_argument("--top_k", type=int, nargs="?", help="top-k value to sample with", default=600)
    parser.add_argument("--temperature", type=float, nargs="?", help="temperature value to sample with", default=1.0)
    parser.add_argument("--top_p", type=float, nargs="?", help="top-p value to sample with", default=0.92)
    parser.add_argument("--chunk_id", type=int, nargs="?", help="chunk id", default=0)
    parser.add_argument("--num_chunks", type=int, nargs="?", help="number of chunks", default=1)
    parser.add_argument("--orig_only", type=str2bool, nargs="?", help="orig only", default=False)
    parser.add_argument("--include_neural_compress", type=str2bool, nargs="?", help="include NC", default=True)
    parser.add_argument("--include_diffpure", type=str2bool, nargs="?", help="include diffpure", default=True)
    parser.add_argument("--wm_method", type=str, nargs="?", help="method", choices=["none", "gentime"])
    parser.add_argument("--wm_seed_strategy", type=str, nargs="?", help="", choices=["fixed", "linear", "spatial"])
    parser.add_argument("--wm_split_strategy", type=str, nargs="?", help="", choices=["rand", "stratifiedrand", "clustering"])
    parser.add_argument("--wm_context_size", type=int, nargs="?", help="context size", default=0)
    parser.add_argument("--wm_delta", type=float, nargs="?", help="wm strength")
    parser.add_argument("--wm_gamma", type=float, nargs="?", help="wm gamma", default=0)
    parser.add_argument("--sync", type=str2bool, default=False)
    parser.add_argument("--syncpath", type=str)
    parser.add_argument("--seed", type=int, nargs="?", help="seed", default=42)
    parser.add_argument("--synthetic", type=str2bool, default=False, help="random-init weights instead of checkpoints")
    parser.add_argument("--synthetic_config", type=str, default="full", choices=["full", "harness"],
                        help="with --synthetic: 'full' = the released architecture; 'harness' = the reduced Taming model of "
                             "tests/golden/harness_vectors.npz (what the reference's own generate.py was run on)")
    parser.add_argument("--noise_device", type=str, default=None, choices=[None, "cpu"],
                        help="'cpu': draw the multinomial noise from the CPU generator (reproduces a CPU run of the reference)")
    parser.add_argument("--augmentations", type=str2bool, default=True,
                        help="run the classic robustness transforms (blur, noise, jpeg, brightness, rotation, flip, crop)")
    return parser


def main():
    sys.path.append(os.getcwd())
    args, _ = get_parser().parse_known_args()
    assert args.outdir, "Output directory is not set"
    assert args.model in ("taming", "rar", "chameleon7b"), f"Model {args.model} not supported"
    assert not args.sync, "--sync (WAM/SyncSeal) is outside the MI355X hot path"
    os.makedirs(args.outdir, exist_ok=True)

    import torch.distributed as dist

    from wmar_amd import harness
    from wmar_amd.models.chameleon_wrapper import ChameleonARMMWrapper
    from wmar_amd.models.rar_wrapper import RarARMMWrapper
    from wmar_amd.models.taming_wrapper import TamingARMMWrapper
    from wmar_amd.utils import synth
    from wmar_amd.utils.utils import update_weights
    from wmar_amd.watermarking.gentime_watermark import GentimeWatermark, SeedStrategy, SplitStrategy

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        dist.init_process_group("nccl")  # RCCL over xGMI
        chunk_id, num_chunks = dist.get_rank(), world
    else:
        chunk_id, num_chunks = args.chunk_id, args.num_chunks
    harness.seed_everything(args.seed, chunk_id)

    device = f"cuda:{local_rank}"
    seed = args.seed + 1000 * chunk_id
    if args.model == "taming":
        if args.synthetic and args.synthetic_config == "harness":
            gcfg, vcfg = synth.GPTConfig(**synth.HARNESS_GPT), synth.VQConfig(**synth.HARNESS_VQ)
            model = TamingARMMWrapper(None, gpt_cfg=gcfg, vq_cfg=vcfg, gpt_state=synth.synth_gpt_state(gcfg, 21, "cpu", 40.0),
                                      vq_state=synth.synth_vq_state(vcfg, 21, "cpu"), device=device,
                                      max_batch=min(args.batch_size, 128))
        elif args.synthetic:
            model = TamingARMMWrapper.synthetic(synth.TAMING_GPT, synth.TAMING_VQ, seed=0, device=device,
                                                max_batch=min(args.batch_size, 128))
        else:
            model = TamingARMMWrapper(args.modelpath, device=device, max_batch=min(args.batch_size, 128))
    elif args.model == "rar":
        if args.synthetic:
            model = RarARMMWrapper.synthetic(device=device, max_batch=min(args.batch_size, 64))
        else:
            model = RarARMMWrapper(args.modelpath, device=device, max_batch=min(args.batch_size, 64))
    else:
        if args.synthetic:
            model = ChameleonARMMWrapper.synthetic(seed=seed, device=device, max_batch=min(args.batch_size, 16))
        else:
            model = ChameleonARMMWrapper(args.modelpath, seed, device=device, max_batch=min(args.batch_size, 16))
    model.noise_device = args.noise_device
    # Patch model: enc and/or dec (the reference's own calls, generate.py:327-332; all three tokenizers expose the handles)
    if args.encoder_ft_ckpt is not None and args.encoder_ft_ckpt != "none":
        update_weights(model.get_image_tokenizer().encoder, args.encoder_ft_ckpt)
    if args.decoder_ft_ckpt is not None and args.decoder_ft_ckpt != "none":
        update_weights(model.get_image_tokenizer().decoder, args.decoder_ft_ckpt)

    if ".txt" in args.conditioning:      # file with prompts (Chameleon): (index, prompt) tuples
        with open(args.conditioning, "r") as f:
            conditionings = [(idx, line.strip()) for idx, line in enumerate(f)]
    else:                                # ImageNet classes
        conditionings = [int(c) for c in args.conditioning.split(",")]
    if args.model == "chameleon7b" and conditionings and isinstance(conditionings[0], int):
        # no prompt file: synthetic prompts (lists of text-token ids derived from the integer) so that --synthetic runs need no tokenizer
        text = model.vocab.text_tokens
        conditionings = [(c, [text[(c * 37 + j * 11) % len(text)] for j in range(12 + c % 5)]) for c in conditionings]
    all_inputs = [c for c in conditionings for _ in range(args.num_samples_per_conditioning)]
    if "chameleon" in args.model or "rar" in args.model:
        assert (args.wm_method in ["none", "gentime"] and args.wm_seed_strategy in ["linear", "fixed"]
                and args.wm_split_strategy == "stratifiedrand"), \
            "Chameleon and RAR models only support none or gentime watermarking with fixed/linear seed and stratifiedrand split"

    vocab_size = model.get_total_vocab_size()
    watermarker = None
    if args.wm_method == "gentime":
        watermarker = GentimeWatermark(model.get_vq(), vocab_size, SeedStrategy(args.wm_seed_strategy),
                                       SplitStrategy(args.wm_split_strategy), args.wm_context_size, args.wm_delta,
                                       args.wm_gamma, model.device)
        if world > 1:  # build the key once, broadcast it over RCCL
            harness.broadcast_key_table(watermarker, device)
    model.set_watermarker(watermarker)

    # evaluation transforms: the classic ones run batched on the GPU; neural codecs and DiffPure are outside this build
    if args.orig_only:
        eval_params = {"metric_names": [], "augmentations": [], "max_roundtrips": 0, "orig_only": True}
    else:
        from wmar_amd.augmentations import AugmentationManager
        if args.include_neural_compress or args.include_diffpure:
            print("WARNING: neural-compression / DiffPure attacks are not part of this build and are skipped", file=sys.stderr)
        augs = AugmentationManager(False, False, load_augs=True).augs if args.augmentations else []
        eval_params = {"metric_names": ["pvalue", "l0", "psnr"], "augmentations": augs, "max_roundtrips": 1,
                       "orig_only": False}
    gen_params = {"batch_size": args.batch_size, "temperature": args.temperature, "top_k": args.top_k,
                  "top_p": args.top_p}
    recs = harness.generate(args.outdir, model, all_inputs, watermarker, eval_params, gen_params, chunk_id=chunk_id,
                            num_chunks=num_chunks)
    if world > 1:
        for r in recs:
            if isinstance(r["conditioning"], tuple):
                r["conditioning"] = r["conditioning"][0]
        recs = harness.gather_records(recs, eval_params, device)      # tensors over RCCL: codes, p-values, l0, psnr
        dist.barrier()
        dist.destroy_process_group()
    if chunk_id == 0 or world == 1:
        with open(os.path.join(args.outdir, "results.json"), "w") as f:
            # the code arrays live in <outdir>/codes/*.npy (30+ records per image x 256-1024 tokens would bloat the json)
            json.dump([{k: (v.tolist() if hasattr(v, "tolist") else v) for k, v in r.items() if k != "codes"} for r in recs], f)
        from wmar_amd.utils.analyzer import summarize
        with open(os.path.join(args.outdir, "summary.json"), "w") as f:      # TPR@1%FPR, mean l0 / PSNR per (method, transform, param)
            json.dump(summarize(recs), f, indent=1)
    print("Done.")


if __name__ == "__main__":
    main()
