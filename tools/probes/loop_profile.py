"""cProfile of the Atari-shaped trainer loop (synthetic env): where the host time of an environment step goes."""
import cProfile, os, pstats, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "i-dqn_amd")]
from experiments.atari.idqn import run
argv = ["-en", "b", "-s", "1", "-ne", "1", "-ntspe", "2000", "-nis", "200", "-rbc", "5000", "-nn", "5", "-at", "cnn",
        "-tuf", "200", "-tsf", "10", "-f", "32", "64", "64", "512", "-horizon", "200", "-bs", "32", "-utd", "4"]
with tempfile.TemporaryDirectory() as d:
    run(argv, save_root=d)  # warm
    pr = cProfile.Profile(); pr.enable(); run(argv, save_root=d); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
