"""`python experiments/lunar_lander/idqn.py -en NAME -s SEED ...` -- counterpart of the reference's entry point of the same
path; the wiring lives in experiments/base/launch.py.  ``env`` defaults to the synthetic stand-in environment."""
import sys

from experiments.base.launch import launch


def run(argvs=sys.argv[1:], env=None, save_root=None):
    return launch("lunar_lander", "idqn", argvs, env=env, save_root=save_root)


if __name__ == "__main__":
    run()
