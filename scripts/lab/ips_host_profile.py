"""cProfile of the host side of one-image-per-call forwards through the shipped IPS detector (hipGraph on): what the CPU does
between the replay launch and the return of the result dicts."""
import os, sys, runpy
os.environ['B1_MODES'] = 'on'
sys.argv = ['ips_image_breakdown.py']
g = runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ips_image_breakdown.py'))
import cProfile, pstats, io, torch
det, clip, meta = g['det'], g['clip'], g['meta']
det.use_graph = True
pr = cProfile.Profile()
pr.enable()
for i in range(40):
    det.forward([clip[i % 8:i % 8 + 1]], [[dict(meta)]], return_loss=False, rescale=True)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22)
print('\n'.join(l[:150] for l in s.getvalue().splitlines()[:40]))
