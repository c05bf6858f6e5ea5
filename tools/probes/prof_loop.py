"""cProfile of the trainer loop on a synthetic environment: python tools/probes/prof_loop.py [atari|lunar_lander]"""
import os, sys, cProfile, pstats, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
which = sys.argv[1] if len(sys.argv) > 1 else "lunar_lander"
if which == "atari":
    from experiments.atari.idqn import run
    argv = ["-en", "b", "-s", "1", "-ne", "1", "-ntspe", "3000", "-nis", "200", "-rbc", "5000", "-nn", "5", "-at", "cnn",
            "-tuf", "200", "-tsf", "10", "-f", "32", "64", "64", "512", "-horizon", "200", "-bs", "32", "-utd", "4"]
else:
    from experiments.lunar_lander.idqn import run
    argv = ["-en", "b", "-s", "1", "-ne", "1", "-ntspe", "3000", "-nis", "200", "-rbc", "10000", "-nn", "3",
            "-tuf", "200", "-tsf", "10", "-f", "100", "100", "-horizon", "200"]
with tempfile.TemporaryDirectory() as d:
    pr = cProfile.Profile(); pr.enable()
    run(argv, save_root=d)
    pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
