import os, sys, cProfile, pstats, tempfile
ROOT = os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
from experiments.lunar_lander.idqn import run
argv = ["-en", "b", "-s", "1", "-ne", "1", "-ntspe", "1500", "-nis", "200", "-rbc", "10000", "-nn", "3",
        "-tuf", "200", "-tsf", "10", "-f", "100", "100", "-horizon", "200"]
with tempfile.TemporaryDirectory() as d:
    pr = cProfile.Profile(); pr.enable()
    run(argv, save_root=d)
    pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
