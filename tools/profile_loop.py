"""cProfile of the Atari-shaped trainer loop (tools/bench_loop.py's second case): where the host time of an environment step goes."""
import cProfile, io, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
import torch
from experiments.atari.idqn import run

argv = ["-en", "b", "-s", "1", "-ne", "1", "-ntspe", "3000", "-nis", "200", "-rbc", "5000", "-nn", "5", "-at", "cnn",
        "-tuf", "200", "-tsf", "10", "-f", "32", "64", "64", "512", "-horizon", "200", "-bs", "32", "-utd", "4"]
with tempfile.TemporaryDirectory() as d:
    run(argv[:7] + ["300"] + argv[8:], save_root=d)  # warm-up (first launches, allocations)
    pr = cProfile.Profile()
    pr.enable()
    run(argv, save_root=d)
    torch.cuda.synchronize()
    pr.disable()
for key in ("tottime", "cumulative"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).strip_dirs().sort_stats(key).print_stats(28)
    print(s.getvalue())
