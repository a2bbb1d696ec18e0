import sys, cProfile, pstats
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/scripts')
import torch, kbench
from openpvsg_amd import unitrack as T
cfg = dict(common=dict(model_type='imagenet50', remove_layers=['layer4'], down_factor=8, infer2D=True, device='cuda'),
           mots=dict(track_buffer=300, conf_thres=0.5, max_mask_area=300, dup_iou_thres=0.15, confirm_iou_thres=0.7,
                     feat_size=[4, 10], use_kalman=True, asso_with_motion=False, motion_lambda=1, motion_gated=False))
vid, outs = kbench.crowd_video(32, 720, 1280, int(sys.argv[1]) if len(sys.argv) > 1 else 16)
model = T.AppearanceModel(cfg).cuda()
T.eval_seq(None, cfg, outs, 126, return_results=True, frames=vid, app_model=model)
pr = cProfile.Profile(); pr.enable()
T.eval_seq(None, cfg, outs, 126, return_results=True, frames=vid, app_model=model)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(45)
