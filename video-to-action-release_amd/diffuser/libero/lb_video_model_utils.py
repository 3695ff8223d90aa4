"""lb_get_video_model_gcp_v2 (diffuser/libero/lb_video_model_utils.py:15-69): Unet_Libero inside GoalGaussianDiffusion
(100 train steps, `timestep` sampling steps, pred_v, cosine, min-SNR, guidance g_w) inside Video_PredModel, weights from
`{ckpts_dir}/model-{milestone}.pt['ema']`.  `allow_random_init=True` (benchmarks / tests only) keeps the seeded default
initialisation when that file does not exist."""
import os
from flowdiffusion.flowdiffusion.goal_diffusion import GoalGaussianDiffusion
from flowdiffusion.flowdiffusion.unet import Unet_Libero
from diffuser.models.video_model import Video_PredModel
from .lb_train_utils import build_text_tower


def lb_get_video_model_gcp_v2(ckpts_dir='../ckpts/metaworld', milestone=24, flow=False, timestep=100, g_w=2.0, sample_per_seq=8,
                              target_size=(128, 128), model_version='luo_128_v0', allow_random_init=False, text_tower=None,
                              **kwargs):
    if model_version != 'luo_128_v0':
        raise NotImplementedError
    unet = Unet_Libero()
    tokenizer, text_encoder = text_tower if text_tower is not None else build_text_tower()
    channels = 3 if not flow else 2
    diffusion = GoalGaussianDiffusion(channels=channels * (sample_per_seq - 1), model=unet, image_size=target_size, timesteps=100,
                                      sampling_timesteps=timestep, loss_type='l2', objective='pred_v', beta_schedule='cosine',
                                      min_snr_loss_weight=True, guidance_weight=g_w)
    video_model = Video_PredModel(diffusion, tokenizer, text_encoder, single_img_channels=channels, results_folder=ckpts_dir)
    if allow_random_init and not os.path.exists(os.path.join(str(ckpts_dir), f"model-{milestone}.pt")):
        print(f"[ lb_video_model_utils ] no checkpoint under {ckpts_dir}: keeping the random initialisation")
    else:
        video_model.load_trained_model(milestone)
    video_model.requires_grad_(False)
    video_model.eval()
    return video_model
