"""cProfile of one rank's routed pass of the 8-rank simulation (host side: where do the gaps between launches go?)."""
import os, sys, cProfile, pstats, io
os.environ['SIM_ONLY'] = 'routed8'
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'exp_multi_rank_sim.py')).read()
# run the simulation script up to its timing loop, then profile 50 steps
src = src.replace("    for _ in range(3):\n        kept, keep, mine = step()", "    for _ in range(5):\n        kept, keep, mine = step()\n    import torch as _t; _t.cuda.synchronize()\n    pr = cProfile.Profile(); pr.enable()\n    for _ in range(50):\n        step()\n    _t.cuda.synchronize(); pr.disable()\n    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(22); print(s.getvalue())")
exec(compile(src.replace("os.path.dirname(os.path.dirname(os.path.abspath(__file__)))", repr(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))), 'sim', 'exec'))
