"""Entry point with the flow of the reference's scripts/train_libero_dp.py:36-162 (dataset -> video model -> policy -> trainer ->
forward/backward smoke test -> train) on the MI355X path.  One process per GPU:

    python scripts/train_libero_dp.py --config config/libero/lb_tk8_65to72.py                      # real data + checkpoints
    python scripts/train_libero_dp.py --synthetic --n_train_steps 400 --init_rand_steps 100         # no simulator / files needed
    torchrun --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 scripts/train_libero_dp.py ...  # data parallel over RCCL

`--synthetic` swaps the three external inputs for in-tree stand-ins: the Libero simulator (toy kinematic env list), the
random-action HDF5 (episodes rolled in that env) and the video checkpoint / CLIP weights (seeded random init, token-hash text
encoder).  Everything between them -- replay store, policy step, video sampler, rollouts, checkpoints -- is the product path."""
import argparse
import importlib.util
import os
import random
import sys
from types import SimpleNamespace

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np      # noqa: E402
import torch            # noqa: E402


def load_config(path):
    spec = importlib.util.spec_from_file_location("v2a_config", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.base


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default=os.path.join(os.path.dirname(HERE), "config/libero/lb_tk8_65to72.py"))
    ap.add_argument("--synthetic", action="store_true")
    ap.add_argument("--savepath", default="logs/libero/diffusion/run0")
    ap.add_argument("--n_train_steps", type=int, default=None)
    ap.add_argument("--init_rand_steps", type=int, default=None)
    ap.add_argument("--video_steps", type=int, default=None, help="sampling steps of the video model (config: timestep)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--precision", choices=["fp32", "bf16"], default="fp32")
    cli = ap.parse_args()

    import v2a_hip                                   # raises if libv2a_hip.so is missing: no silent fallback
    v2a_hip.set_precision(cli.precision)
    from diffuser.libero import _host_utils as utils
    from diffuser.libero.lb_video_model_utils import lb_get_video_model_gcp_v2
    from diffuser.libero.lb_train_utils import LB_Init_Trainer
    from diffuser.diffusion_policy import Init_Diffusion_Policy

    base = load_config(cli.config)
    args = SimpleNamespace(dataset=base['dataset'], savepath=cli.savepath, **base['diffusion'])
    rank = int(os.environ.get("RANK", "0"))
    torch.manual_seed(cli.seed + rank); np.random.seed(cli.seed + rank); random.seed(cli.seed + rank)
    if cli.synthetic:
        os.environ.setdefault("V2A_TEXT_ENCODER", "hash")
        args.dataset = "synthetic-8tk-65to72"
        args.trainer_dict['randsam_filename'] = 'synthetic:500'
        args.vid_diffusion['allow_random_init'] = True
        args.model_yl_path = None
    if cli.n_train_steps is not None:
        args.n_train_steps = cli.n_train_steps
    if cli.init_rand_steps is not None:
        args.trainer_dict['init_rand_steps'] = cli.init_rand_steps
    if cli.video_steps is not None:
        args.vid_diffusion['timestep'] = cli.video_steps
    args.trainer_dict['seed'] = cli.seed

    dataset_config = utils.Config(args.loader, savepath=(args.savepath, 'dataset_config.pkl'), env=args.dataset,
                                  target_size=args.input_img_size, dataset_config=args.dataset_config)
    dataset = dataset_config()

    assert tuple(args.input_img_size) == (128, 128)
    args.vid_diffusion['target_size'] = args.input_img_size
    init_tr = LB_Init_Trainer(args)
    video_model = lb_get_video_model_gcp_v2(text_tower=(init_tr.tokenizer, init_tr.text_encoder), **args.vid_diffusion)
    init_diff_policy = Init_Diffusion_Policy(args)

    trainer_dict = args.trainer_dict
    trainer_config = utils.Config(
        init_tr.trainer_cls, savepath=None, channels=3, train_batch_size=trainer_dict['batch_size'],
        video_batch_size=trainer_dict.get('batch_size_v', 4), valid_batch_size=1,
        gradient_accumulate_every=args.gradient_accumulate_every, augment_horizontal_flip=None,
        train_num_steps=args.n_train_steps, opt_params=args.opt_params, ema_params=args.ema_params,
        render_img_size=args.render_img_size, input_img_size=args.input_img_size, sample_freq=args.sample_freq,
        save_freq=args.save_freq, label_freq=int(args.n_train_steps // args.n_saves), log_freq=args.log_freq,
        n_samples=args.n_samples, results_folder=args.savepath, trainer_dict=trainer_dict)
    trainer = trainer_config(init_diff_policy=init_diff_policy, video_model=video_model, tokenizer=init_tr.tokenizer,
                             text_encoder=init_tr.text_encoder, train_set=dataset, valid_set=dataset)
    if getattr(args, 'do_train_resume', False):
        trainer.load(utils.get_latest_epoch((trainer.results_folder,)))

    gcp_model = init_diff_policy.diffusion_policy
    utils.report_parameters(gcp_model)
    print('Testing forward...', end=' ', flush=True)
    device = next(gcp_model.parameters()).device
    img1, img2, _, act = dataset.sample_random_tensor(1, trainer_dict['model_act_horizon'], device)
    loss = gcp_model.compute_loss(trainer.to_batch_dict(img1, img2, act))
    loss.backward()
    trainer.opt.zero_grad()
    print('ok')

    trainer.train()


if __name__ == "__main__":
    main()
