"""`python experiments/atari/iiqn.py -en NAME -s SEED -at cnn -f 32 64 64 512 -nn 5 -nq 32 ...` -- trainer entry point of
the i-IQN extension (BASELINE config 3; the reference has no such script: its README points at another repository).
Same flags as atari/idqn.py plus ``-nq`` (quantile fractions per sample and pass); the wiring lives in
experiments/base/launch.py.  ``env`` defaults to the synthetic stand-in environment."""
import sys

from experiments.base.launch import launch


def run(argvs=sys.argv[1:], env=None, save_root=None):
    return launch("atari", "iiqn", argvs, env=env, save_root=save_root)


if __name__ == "__main__":
    run()
