"""Entry point with the reference's CLI (run.py:64-71): `python run.py --config_file hparams_base.yaml`.

Differences forced by the offline environment (DESIGN.md "Out of scope"): the Peppa dataset and
moviepy are unavailable, so training runs on synthetic clips of the configured shape.  The shipped
yaml files say `pretrained: true`; the Kinetics / fairseq weights cannot be downloaded, so either
pass `--random_init` (same architectures from random init; the saved config then records
`pretrained: false`) or point `mi355x: {video_weights, audio_weights}` in the yaml at local
state_dict files -- otherwise the run stops with an error instead of silently deviating.
The loop is always peppa_amd.trainer.Trainer (the subset of `pl.Trainer` the reference uses,
run.py:56-62); the Trainer flags it implements are accepted below, other `pl.Trainer` flags are
reported and ignored.  Top-level config keys can be overridden from the command line exactly as
in the reference (run.py:25-27)."""
import logging
import os
from argparse import ArgumentParser

import torch
import yaml

import pig.models
from pig.execution import default_config
from peppa_amd.checkpoint import ModelCheckpoint
from peppa_amd.trainer import SyntheticPigData, Trainer


def get_git_commit():
    try:  # GitPython is optional; never fail outside a git checkout
        import git
        return git.Repo(os.getcwd()).head.reference.commit.hexsha
    except Exception:
        return None


def main(args):
    logging.getLogger().setLevel(logging.INFO)
    logging.basicConfig()
    config = default_config if args.config_file is None else yaml.safe_load(open(args.config_file))
    for key, value in vars(args).items():
        if key in config and value is not None:
            config[key] = value
    config['git_commit'] = get_git_commit()
    if args.deterministic:
        config.setdefault('mi355x', {})
        config['mi355x'] = dict(config['mi355x'] or {}, deterministic=True)
    if args.random_init:
        config['video']['pretrained'] = False
        config['audio']['pretrained'] = False
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        from peppa_amd.video import ensure_streams
        ensure_streams(f"cuda:{local_rank}")            # the step's side streams before RCCL creates its own
        torch.distributed.init_process_group("nccl")    # lazy communicator: an eager one (device_id=) slows every kernel
    batch_size = config['data']['train']['batch_size']
    data = SyntheticPigData(config['data'], frames=args.frames, size=args.size, samples=args.samples,
                            steps_per_epoch=args.limit_train_batches or 100, device=f"cuda:{local_rank}",
                            val_batches=max(2, -(-100 // batch_size)))   # resampled_recall draws 100 clips (pig/metrics.py:55-56)
    net = pig.models.PeppaPig(config).to(f"cuda:{local_rank}")
    targs = dict(config['training']['trainer_args'])
    for key in ('accumulate_grad_batches', 'precision'):          # Trainer flags on the command line win (run.py:59-61)
        if getattr(args, key, None) is not None:
            targs[key] = getattr(args, key)
    if str(targs.get('precision', 16)) not in ('16', 'bf16', 'fp16'):
        raise SystemExit(f"precision {targs['precision']}: the HIP path computes in 16-bit (bf16 or fp16 operands, fp32 accumulate)")
    # the reference's two checkpoint callbacks (run.py:32-55): best epoch by narration recall@10 and by triplet accuracy
    callbacks = [ModelCheckpoint(monitor=monitor, mode='max', save_last=True, save_top_k=1,
                                 filename="{epoch}-{" + monitor + ":.2f}")
                 for monitor in ('valnarr_rec_fixed', 'valnarr_triplet')] if args.default_root_dir else []
    trainer = Trainer(accumulate_grad_batches=targs.get('accumulate_grad_batches', 1),
                      limit_train_batches=args.limit_train_batches, limit_val_batches=args.limit_val_batches,
                      max_epochs=args.max_epochs, max_steps=args.max_steps, callbacks=callbacks,
                      default_root_dir=args.default_root_dir, resume_from_checkpoint=args.resume_from_checkpoint,
                      log_every=args.log_every_n_steps, max_time_s=2 * 24 * 3600, precision=targs.get('precision'))
    trainer.fit(net, data)
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    parser = ArgumentParser()
    parser.add_argument("--config_file", help="Configuration file (YAML)", default=None)
    parser.add_argument("--limit_train_batches", type=int, default=None)
    parser.add_argument("--limit_val_batches", type=int, default=None)
    parser.add_argument("--margin", type=float, default=None)
    parser.add_argument("--max_epochs", type=int, default=1)
    parser.add_argument("--default_root_dir", default=None,
                        help="write Lightning-format checkpoints under {dir}/checkpoints after each validation pass")
    parser.add_argument("--max_steps", type=int, default=None)
    parser.add_argument("--accumulate_grad_batches", type=int, default=None)
    parser.add_argument("--precision", default=None,
                        help="16 / bf16: bf16 operands (default on MI355X); fp16: IEEE half + dynamic loss scaling, as the "
                             "reference's AMP runs")
    parser.add_argument("--gpus", default=None, help="ignored: one process per GPU (torchrun sets WORLD_SIZE / LOCAL_RANK)")
    parser.add_argument("--resume_from_checkpoint", default=None)
    parser.add_argument("--log_every_n_steps", type=int, default=10)
    parser.add_argument("--random_init", action="store_true",
                        help="pretrained weights cannot be downloaded offline: build the same architectures from random "
                             "init (sets video.pretrained / audio.pretrained to false in the saved config)")
    parser.add_argument("--deterministic", action="store_true",
                        help="bitwise-reproducible steps (pp_set_option('deterministic', 1): ordered reductions, slower)")
    parser.add_argument("--frames", type=int, default=16)
    parser.add_argument("--size", type=int, default=112)
    parser.add_argument("--samples", type=int, default=36800)
    args, unknown = parser.parse_known_args()
    if unknown:
        logging.warning("pl.Trainer flags not implemented by the built-in loop, ignored: %s", " ".join(unknown))
    main(args)
