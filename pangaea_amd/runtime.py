"""Process-level helpers behind the reference's ``utils`` names (src/utils.py): the sampler that draws row indices with
numpy's global generator (utils.py:9-21), checkpoint-on-improvement early stopping (utils.py:24-50), the external-tool
runner (utils.py:67-79) and the seeding + logging set-up (utils.py:82-103).  ``pangaea_amd.utils`` re-exports them under
the reference's module name."""
from __future__ import annotations

import contextlib
import logging
import os
import subprocess
import sys
import threading

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
    """Call with (validation loss, model) after every validation pass.  A loss at least ``delta`` below the best one seen
    saves the model's ``state_dict`` to ``path`` and resets the patience; ``patience`` passes in a row without such an
    improvement set ``early_stop``.  ``best_score`` (= minus the best loss), ``counter`` and ``val_loss_min`` (the loss of
    the last checkpoint) are kept as attributes because callers of the reference's class read them."""

    def __init__(self, patience: int = 7, delta: float = 0, path: str = "checkpoint.pt"):
        self.patience, self.delta, self.path = patience, delta, path
        self.best_score = None
        self.val_loss_min = float("inf")
        self.counter = 0
        self.early_stop = False

    def _improved(self, val_loss) -> bool:
        return self.best_score is None or -val_loss >= self.best_score + self.delta

    def __call__(self, val_loss, model) -> None:
        if self._improved(val_loss):
            self.best_score, self.counter = -val_loss, 0
            self.save_checkpoint(val_loss, model)
        else:
            self.counter += 1
            self.early_stop = self.early_stop or self.counter >= self.patience

    def save_checkpoint(self, val_loss, model) -> None:
        torch.save(model.state_dict(), self.path)
        self.val_loss_min = val_loss


def run_cmd(command, log_file=None) -> None:
    """run an external tool with its stdout swallowed and its stderr appended to ``log_file`` (or dropped); a non-zero exit
    status ends the whole program with status 1, as the reference's pipeline expects of a failed step"""
    line = " ".join(map(str, command))
    logging.info("running: %s", line)
    with contextlib.ExitStack() as stack:
        err = stack.enter_context(open(log_file, "a")) if log_file else subprocess.DEVNULL
        done = subprocess.run(command, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=err, text=True)
    if done.returncode != 0:
        logging.error("failed with status %d: %s", done.returncode, line)
        sys.exit(1)
    logging.info("finished: %s", line)


def init_all(seed, threads, logfile, level, outdir, file_log: bool = True) -> None:
    """seed numpy and torch (the samplers and the network initialisation take their randomness from there), bound torch's
    host threads, create the output directory and send log records of ``level`` and above to ``outdir/logfile`` and stderr
    (``file_log`` False: stderr only -- of several ranks sharing one output directory only the first writes the file)"""
    for seeder in (np.random.seed, torch.manual_seed, torch.cuda.manual_seed_all):
        seeder(seed)
    torch.set_num_threads(threads)
    os.makedirs(outdir, exist_ok=True)
    log = logging.getLogger()
    log.setLevel(level)
    layout = logging.Formatter("%(asctime)s (%(levelname)s): %(message)s", "%Y-%m-%d %H:%M:%S")
    for sink in ([logging.FileHandler(os.path.join(outdir, logfile))] if file_log else []) + [logging.StreamHandler()]:
        sink.setLevel(level)
        sink.setFormatter(layout)
        log.addHandler(sink)
    log.info("pangaea (MI355X feature path) starting, output in %s", outdir)


_warm_threads: list = []


def join_warm_blas() -> None:
    """wait for every ``warm_blas`` thread started so far (before a HIP-graph capture: a capture does not tolerate another
    thread of the process putting work on a stream meanwhile)"""
    while _warm_threads:
        _warm_threads.pop().join()


def warm_blas(device, widths=(536, 512, 512, 32), rows: int = 4096):
    """Start the GEMM library's one-time initialisation for ``device`` on a helper thread and return the thread (or None without
    a GPU).  The first matrix product of a process loads hipBLASLt/rocBLAS and their kernel tables: 190 ms on an MI355X box
    with a warm page cache, several times that on a fresh one -- more than the whole feature pass over 10 M read pairs takes,
    and until round 4 it sat in front of the first encode.  The reference's flow is features first, network afterwards
    (/root/reference/src/pangaea.py:70,90): the initialisation runs under the ingest instead.  ``widths`` are the layer widths
    of the encoder the products are warmed for (VAENET: tnf + abundance columns -> 512 -> 512 -> latent); PANGAEA_WARM_BLAS=0
    turns it off."""
    device = torch.device(device)
    if device.type != "cuda" or os.environ.get("PANGAEA_WARM_BLAS", "1") in ("", "0"):
        return None

    def work():
        try:
            with torch.cuda.device(device), torch.no_grad():
                x = torch.zeros((rows, widths[0]), dtype=torch.float32, device=device)
                for a, b in zip(widths[:-1], widths[1:]):
                    x = torch.nn.functional.linear(x, torch.zeros((b, a), dtype=torch.float32, device=device), torch.zeros(b, dtype=torch.float32, device=device))
                torch.cuda.current_stream(device).synchronize()
        except RuntimeError as err:                              # (the real product will report what is wrong, in its own place)
            logging.debug("warm_blas: %s", err)

    # (not a daemon: a process that ends while the library is still initialising would kill the thread inside the driver --
    # glibc aborts on the mutex it holds; the interpreter waits for the thread instead)
    t = threading.Thread(target=work, name="pangaea-warm-blas", daemon=False)
    t.start()
    _warm_threads.append(t)
    return t
