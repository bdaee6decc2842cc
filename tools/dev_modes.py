"""Parity of every fit-kernel shape (k2b_fit_config.debug_launch_shape) against the golden cases + timing."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, torch
from tests import helpers as H
for mode in ('split', 'split_paired', 'paired'):
    worst = 0.0
    for case in H.WORLD_CASES:
        d = H.load_case(case)
        out = H.native_fit(d, shape=mode)
        e = max(np.abs(out[k].cpu().numpy() - d['out_' + k]).max() for k in ('global_orient', 'body_pose', 'betas', 'transl'))
        worst = max(worst, e)
    print(f'mode {mode}: worst final |param diff| vs reference over {len(H.WORLD_CASES)} cases: {worst:.2e}')
