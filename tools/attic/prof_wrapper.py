"""Where does the Python wrapper of rnms spend its time? (cProfile over 2000 calls, n = 2048)"""
import os, sys, cProfile, pstats, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle
from aidet_amd import rnms
d = torch.from_numpy(oracle.synth_boxes(2048, 2851, 5)).cuda()
for _ in range(20): rnms(d, 0.5)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(2000): rnms(d, 0.5)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr).sort_stats('cumulative')
st.print_stats(18)
