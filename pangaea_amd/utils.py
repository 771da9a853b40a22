"""Counterparts of /root/reference/src/utils.py used on the hot path: the weighted sampler (utils.py:9-21), early
stopping with checkpoint-on-improve (utils.py:24-50), ``run_cmd`` (utils.py:67-79) and ``init_all`` (utils.py:82-103)."""
from __future__ import annotations

import logging
import os

import numpy as np
import torch
from torch.utils.data.sampler import WeightedRandomSampler


class CustomWeightedRandomSampler(WeightedRandomSampler):
    """``np.random.choice`` over the row weights -- with or without replacement -- from numpy's global generator,
    which ``init_all`` seeds; the draw is therefore the reference's draw for the same seed."""

    def __iter__(self):
        w = self.weights.numpy()
        picks = np.random.choice(len(w), size=self.num_samples, p=w / w.sum(), replace=self.replacement)
        return iter(picks.tolist())


class EarlyStopping:
    def __init__(self, patience=7, delta=0, path="checkpoint.pt"):
        self.patience, self.delta, self.path = patience, delta, path
        self.counter = 0
        self.best_score = None
        self.early_stop = False
        self.val_loss_min = np.inf

    def __call__(self, val_loss, model):
        score = -val_loss
        if self.best_score is not None and score < self.best_score + self.delta:
            self.counter += 1
            self.early_stop = self.counter >= self.patience
            return
        self.best_score = score
        self.save_checkpoint(val_loss, model)
        self.counter = 0

    def save_checkpoint(self, val_loss, model):
        torch.save(model.state_dict(), self.path)
        self.val_loss_min = val_loss


def run_cmd(command, log_file=None):
    """run an external tool, exit the program on failure (utils.py:67-79)"""
    import subprocess
    import sys
    log_pipe = subprocess.DEVNULL if not log_file else open(log_file, "a")
    logging.info("command started: " + " ".join(command))
    ret = subprocess.run(command, stdout=subprocess.PIPE, stderr=log_pipe, stdin=subprocess.PIPE, text=True)
    if ret.returncode:
        logging.error("command failed: " + " ".join(command))
        sys.exit(1)
    logging.info("command completed: " + " ".join(command))


def init_all(seed, threads, logfile, level, outdir):
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    torch.set_num_threads(threads)
    os.makedirs(outdir, exist_ok=True)
    root = logging.getLogger()
    root.setLevel(level)
    fmt = logging.Formatter("%(asctime)s (%(levelname)s): %(message)s", "%Y-%m-%d %H:%M:%S")
    for h in (logging.FileHandler(os.path.join(outdir, logfile)), logging.StreamHandler()):
        h.setLevel(level)
        h.setFormatter(fmt)
        root.addHandler(h)
    root.info("program start up")
