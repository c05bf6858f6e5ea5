"""Logging / persistence glue of the trainer (reference ``experiments/base/utils.py:12-134``), reduced to what the
hot path's callers need: the flag set (names and defaults verbatim from ``experiments/base/parser_argument.py``),
a logger with wandb's ``.log`` and the per-epoch JSON + pickle dump (``utils.py:123-134``)."""
import argparse
import json
import os
import pickle


class NullLogger:
    def __init__(self):
        self.records = []

    def log(self, record):
        self.records.append(dict(record))


def base_parser(algo_name: str) -> argparse.ArgumentParser:
    ap = argparse.ArgumentParser(f"Train {algo_name}.")
    ap.add_argument("-en", "--experiment_name", type=str, required=True)
    ap.add_argument("-s", "--seed", type=int, required=True)
    ap.add_argument("-dw", "--disable_wandb", default=False, action="store_true")
    ap.add_argument("-f", "--features", nargs="*", type=int, default=[100, 100])
    ap.add_argument("-rbc", "--replay_buffer_capacity", type=int, default=10_000)
    ap.add_argument("-bs", "--batch_size", type=int, default=32)
    ap.add_argument("-n", "--update_horizon", type=int, default=1)
    ap.add_argument("-gamma", "--gamma", type=float, default=0.99)
    ap.add_argument("-lr", "--learning_rate", type=float, default=3e-4)
    ap.add_argument("-horizon", "--horizon", type=int, default=1000)
    ap.add_argument("-at", "--architecture_type", type=str, default="fc", choices=["cnn", "impala", "fc"])
    ap.add_argument("-ne", "--n_epochs", type=int, default=50)
    ap.add_argument("-ntspe", "--n_training_steps_per_epoch", type=int, default=10_000)
    ap.add_argument("-utd", "--update_to_data", type=float, default=1)
    ap.add_argument("-nis", "--n_initial_samples", type=int, default=1_000)
    ap.add_argument("-ee", "--epsilon_end", type=float, default=0.01)
    ap.add_argument("-ed", "--epsilon_duration", type=float, default=1_000)
    ap.add_argument("-tuf", "--target_update_frequency", type=int, default=200)
    if algo_name in ("idqn", "iiqn"):
        ap.add_argument("-nn", "--n_networks", type=int, default=3)
        ap.add_argument("-tsf", "--target_sync_frequency", type=int, default=10)
    if algo_name == "iiqn":  # extension (the reference has no quantile agent): fractions per sample and pass
        ap.add_argument("-nq", "--n_quantiles", type=int, default=32)
    return ap


def prepare_logs(env_name: str, algo_name: str, argvs, save_root=None):
    p = vars(base_parser(algo_name).parse_args(argvs))
    p["env_name"], p["algo_name"] = env_name, algo_name
    root = save_root or os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", env_name, "exp_output")
    p["save_path"] = os.path.join(root, p["experiment_name"], algo_name)
    p["wandb"] = NullLogger()
    return p


def save_data(p: dict, episode_returns: list, episode_lengths: list, model):
    os.makedirs(os.path.join(p["save_path"], "episode_returns_and_lengths"), exist_ok=True)
    os.makedirs(os.path.join(p["save_path"], "models"), exist_ok=True)
    with open(os.path.join(p["save_path"], f"episode_returns_and_lengths/{p['seed']}.json"), "w") as f:
        json.dump({"episode_lengths": episode_lengths, "episode_returns": episode_returns}, f, indent=4)
    with open(os.path.join(p["save_path"], f"models/{p['seed']}"), "wb") as f:
        pickle.dump(model, f)
