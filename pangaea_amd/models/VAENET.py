"""``VAENET`` / ``VaritionalAutoEncoder`` -- device-resident counterpart of /root/reference/src/models/VAENET.py.

Interface kept: ``VAENET(abd_dim, tnf_dim, latent_size, num_classes, epochs, cuda, num_gpus, lr, dropout, alpha,
w_kl, weight_decay).train(train_loader, val_loader, dataloader, model_path, patience)`` writes
``train_model.pk`` (state_dict with the reference's keys ``encoder.{0,1,4,5}``, ``l_mu``, ``l_sigma``,
``decoder.{0,1,4,5}``, ``output``), ``latent.npz``, ``barcodes.npz`` and ``model_finished`` (VAENET.py:35,128-149).

Behaviour reproduced on purpose:
  * ``nn.LeakyReLU(True)`` -- negative_slope=True == 1.0, i.e. the identity (VAENET.py:205,217);
  * hidden sizes [512, 512] and latent 32 whatever ``latent_size`` says; it only scales ``w_kl`` (VAENET.py:20,25).
Changed on purpose: the network really lives on the GPU (the reference unwraps DataParallel and draws epsilon on
the CPU, VAENET.py:26-29,227, so its ``-g`` flag cannot work); ``encode`` embeds a whole ``Data`` set from its
device copies in dataset order instead of going through a shuffling DataLoader -- ``barcodes.npz`` carries the
order, as in the reference.
"""
from __future__ import annotations

import logging
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

from ..runtime import join_warm_blas
from ..utils import EarlyStopping


class VaritionalAutoEncoder(nn.Module):
    def __init__(self, input_abd_size, input_tnf_size, hidden_sizes=(512, 512), latent_size=32, dropout=0.2):
        super().__init__()
        hidden_sizes = list(hidden_sizes)
        self.abd_size, self.tnf_size = input_abd_size, input_tnf_size
        self.input_size = input_abd_size + input_tnf_size

        def stack(sizes):
            layers = []
            for n_in, n_out in zip(sizes[:-1], sizes[1:]):
                layers += [nn.Linear(n_in, n_out), nn.BatchNorm1d(n_out), nn.LeakyReLU(True), nn.Dropout(dropout)]
            return nn.Sequential(*layers)

        self.encoder = stack([self.input_size] + hidden_sizes)
        self.l_mu = nn.Linear(hidden_sizes[-1], latent_size)
        self.l_sigma = nn.Linear(hidden_sizes[-1], latent_size)
        self.softplus = nn.Softplus()
        self.decoder = stack([latent_size] + hidden_sizes[::-1])
        self.output = nn.Linear(hidden_sizes[0], self.input_size)

    def calcu_latent(self, abd, tnf, epsilon=None):
        hidden = self.encoder(torch.cat((abd, tnf), 1))
        mu = self.l_mu(hidden)
        logsigma = self.softplus(self.l_sigma(hidden))
        if epsilon is None:
            epsilon = torch.randn(mu.size(0), mu.size(1), device=mu.device)
        latent = mu + epsilon * torch.exp(logsigma / 2)
        return mu, logsigma, latent

    def emebdding(self, abd, tnf):          # (sic) the reference's spelling, VAENET.py:232
        return self.l_mu(self.encoder(torch.cat((abd, tnf), 1)))

    def forward(self, abd, tnf, epsilon=None):
        mu, logsigma, latent = self.calcu_latent(abd, tnf, epsilon)
        out = self.output(self.decoder(latent))
        return {
            "abd": abd, "tnf": tnf,
            "abd_rec": F.softmax(out.narrow(1, 0, self.abd_size), dim=1),
            "tnf_rec": F.softmax(out.narrow(1, self.abd_size, self.tnf_size), dim=1),
            "mu": mu, "logsigma": logsigma,
        }


class VAENET:
    def __init__(self, abd_dim, tnf_dim, latent_size, num_classes, epochs, cuda, num_gpus, lr, dropout, alpha, w_kl,
                 weight_decay):
        self.num_epochs = epochs
        self.num_classes = num_classes
        self.latent_size = latent_size
        self.input_size = abd_dim + tnf_dim
        self.learning_rate = lr
        self.weight_decay = weight_decay
        self.w_kl = w_kl * 100 / latent_size
        self.wa = alpha * 100 / np.log(abd_dim)
        self.wt = (1 - alpha) * 100 / np.log(tnf_dim)
        self.cuda = bool(cuda)
        self.eps = 1e-9
        self.device = torch.device("cuda", torch.cuda.current_device()) if self.cuda else torch.device("cpu")
        self.network = VaritionalAutoEncoder(abd_dim, tnf_dim, dropout=dropout).to(self.device)

    # ------------------------------------------------------------------ loss (VAENET.py:161-184)

    def reconstruction_loss(self, real, predicted):
        return -(torch.log(predicted + self.eps) * real).sum(-1).mean()

    def unlabeled_loss(self, out_net):
        mu, logsigma = out_net["mu"], out_net["logsigma"]
        loss_abd = self.reconstruction_loss(out_net["abd"], out_net["abd_rec"])
        loss_tnf = self.reconstruction_loss(out_net["tnf"], out_net["tnf_rec"])
        loss_kl = -0.5 * (1 + logsigma - mu.pow(2) - logsigma.exp()).sum(dim=1).mean()
        total = self.wa * loss_abd + self.wt * loss_tnf + self.w_kl * loss_kl
        return {"total": total, "abd_rec": loss_abd, "tnf_rec": loss_tnf, "kl_loss": loss_kl}

    # ------------------------------------------------------------------ encode (VAENET.py:126-149)

    @torch.no_grad()
    def encode(self, data, batch_rows: int = 1 << 16) -> torch.Tensor:
        """mu for every row of a ``pangaea_amd.data.Data`` set, in dataset order, on the device"""
        self.network.eval()
        abd, tnf = data.abd_dev.to(self.device), data.tnf_dev.to(self.device)
        out = torch.empty((abd.shape[0], self.network.l_mu.out_features), dtype=torch.float32, device=self.device)
        for a in range(0, abd.shape[0], batch_rows):
            out[a:a + batch_rows] = self.network.emebdding(abd[a:a + batch_rows], tnf[a:a + batch_rows])
        return out

    # ------------------------------------------------------------------ hipGraph capture of the two launch-bound loops
    #
    # A training step is ~90 small kernels (two 512-wide hidden layers on 2048 rows: forward, backward, Adam), a
    # validation forward ~25: the GPU finishes them faster than the host can launch them.  Once the first batches have
    # run eagerly, the step is captured into a HIP graph with static input buffers and replayed per batch (same kernels,
    # same order; BatchNorm statistics, dropout masks and epsilon advance inside the graph).  Batches of another size
    # (the last one of an epoch) run eagerly.  PG_TRAIN_GRAPH=0 turns the capture off.

    def _graph_ok(self) -> bool:
        return self.cuda and os.environ.get("PG_TRAIN_GRAPH", "1") != "0"

    def _capture_train_step(self, opt, abd, tnf, side):
        join_warm_blas()                                # (runtime.warm_blas: its helper thread must not launch during a capture)
        static_abd, static_tnf = abd.clone(), tnf.clone()
        self.network.train()
        opt.zero_grad(set_to_none=True)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):      # the stream the eager warm-up steps ran on (their AccumulateGrad nodes)
            losses = self.unlabeled_loss(self.network(static_abd, static_tnf))
            losses["total"].backward()
            opt.step()
            vec = torch.stack([losses[k].detach() for k in ("total", "abd_rec", "tnf_rec", "kl_loss")])

        def step(a, t):
            static_abd.copy_(a); static_tnf.copy_(t)
            graph.replay()
            v = vec.clone()
            return {"total": v[0], "abd_rec": v[1], "tnf_rec": v[2], "kl_loss": v[3]}
        step.rows = int(abd.shape[0])
        step.keep = (graph, static_abd, static_tnf, vec)
        return step

    def _capture_val_step(self, abd, tnf):
        join_warm_blas()
        static_abd, static_tnf = abd.clone(), tnf.clone()
        self.network.eval()
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            total = self.unlabeled_loss(self.network(static_abd, static_tnf))["total"]

        def step(a, t):
            static_abd.copy_(a); static_tnf.copy_(t)
            graph.replay()
            return total.clone()
        step.rows = int(abd.shape[0])
        step.keep = (graph, static_abd, static_tnf, total)
        return step

    def _to_dev(self, batch):
        return torch.as_tensor(batch["abd"]).to(self.device), torch.as_tensor(batch["tnf"]).to(self.device)

    @torch.no_grad()
    def _validate(self, loader) -> float:
        self.network.eval()
        losses = []                     # kept on the device: one host sync per validation pass, not one per batch
        seen = 0
        for batch in loader:
            abd, tnf = self._to_dev(batch)
            graphed = getattr(self, "_val_step", None)
            if graphed is not None and graphed.rows == abd.shape[0]:
                losses.append(graphed(abd, tnf))
                continue
            losses.append(self.unlabeled_loss(self.network(abd, tnf))["total"])
            seen += 1
            if graphed is None and seen == 3 and self._graph_ok():      # warmed up: capture the forward for the rest
                torch.cuda.synchronize()
                self._val_step = self._capture_val_step(abd, tnf)
        return float(torch.stack(losses).double().mean().item()) if losses else float("nan")

    # ------------------------------------------------------------------ train (VAENET.py:31-149)

    @staticmethod
    def write_latent(model_path: str, embedding, barcodes) -> None:
        """``latent.npz`` / ``barcodes.npz`` / ``model_finished`` as the reference's train() leaves them (VAENET.py:128-149)"""
        np.savez(os.path.join(model_path, "barcodes.npz"), barcodes)
        np.savez(os.path.join(model_path, "latent.npz"), embedding)
        with open(os.path.join(model_path, "model_finished"), "w") as f:
            f.write("model finished")

    def train(self, train_loader, val_loader, dataloader, model_path, patience, encode: bool = True):
        if not os.path.isdir(model_path):
            raise Exception("model path not exist")
        train_model = os.path.join(model_path, "train_model.pk")
        early = EarlyStopping(patience=patience, delta=1e-6, path=train_model)
        if not os.path.exists(train_model):
            logging.info("train start")
            opt = torch.optim.Adam(self.network.parameters(), lr=self.learning_rate, weight_decay=self.weight_decay,
                                   capturable=self._graph_ok())
            self._val_step = None
            train_step, eager_steps, side = None, 0, None
            hist = {"total": [], "abd_rec": [], "tnf_rec": [], "kl_loss": []}

            def report(epoch, batch, val):
                # the running losses stay on the device until they are printed (the reference calls .item() four times per
                # batch, VAENET.py:70-76: on a GPU that is four pipeline drains per step)
                avg = {k: float(torch.stack(v).double().mean().item()) if v else float("nan") for k, v in hist.items()}
                logging.info(
                    f"epoch {epoch}/{self.num_epochs} batch {batch + 1}/{len(train_loader)}: train {avg['total']:.8f} "
                    f"abd {avg['abd_rec']:.8f} tnf {avg['tnf_rec']:.8f} "
                    f"kl {avg['kl_loss']:.8f} | test {val:.8f}")
                for v in hist.values():
                    v.clear()

            for epoch in range(1, self.num_epochs + 1):
                batch = -1
                for batch, data in enumerate(train_loader):
                    self.network.train()
                    abd, tnf = self._to_dev(data)
                    if train_step is not None and train_step.rows == abd.shape[0]:
                        losses = train_step(abd, tnf)
                        for key in hist:
                            hist[key].append(losses[key])
                    elif self._graph_ok() and train_step is None:
                        # warm-up steps (real ones) on the side stream the graph will be captured on
                        if side is None:
                            side = torch.cuda.Stream()
                        side.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side):
                            opt.zero_grad(set_to_none=True)
                            losses = self.unlabeled_loss(self.network(abd, tnf))
                            for key in hist:
                                hist[key].append(losses[key].detach())
                            losses["total"].backward()
                            opt.step()
                        torch.cuda.current_stream().wait_stream(side)
                        del losses                      # nothing may keep the eager autograd graph alive across the capture
                        eager_steps += 1
                        if eager_steps == 3:
                            torch.cuda.synchronize()
                            train_step = self._capture_train_step(opt, abd, tnf, side)
                    else:
                        opt.zero_grad()
                        losses = self.unlabeled_loss(self.network(abd, tnf))
                        for key in hist:
                            hist[key].append(losses[key].detach())
                        losses["total"].backward()
                        opt.step()
                    if (batch + 1) % 100 == 0:          # validation + early stopping every 100 batches
                        val = self._validate(val_loader)
                        report(epoch, batch, val)
                        early(val, self.network)
                    if early.early_stop:
                        logging.info("early stop triggered")
                        break
                if early.early_stop:
                    logging.info("early stop triggered")
                    break
                val = self._validate(val_loader)
                report(epoch, batch, val)
                if len(train_loader) % 100 != 0:
                    early(val, self.network)
                    if early.early_stop:
                        logging.info("early stop triggered")
                        break
            if not os.path.exists(train_model):
                torch.save(self.network.state_dict(), train_model)
        else:
            logging.info("trainning model already saved")

        if not encode:                  # several ranks: the rows are encoded where they lie (pangaea.run), the files follow there
            return
        latent_path = os.path.join(model_path, "latent.npz")
        barcodes_path = os.path.join(model_path, "barcodes.npz")
        if not os.path.exists(latent_path) or not os.path.exists(barcodes_path):
            self.network.load_state_dict(torch.load(train_model, map_location=self.device))
            self.network.eval()
            dataset = getattr(dataloader, "dataset", None)
            if dataset is not None and hasattr(dataset, "abd_dev"):
                embedding = self.encode(dataset).cpu().numpy()
                barcodes = list(dataset.bc)
            else:
                chunks, barcodes = [], []
                with torch.no_grad():
                    for data in dataloader:
                        abd, tnf = self._to_dev(data)
                        chunks.append(self.network.emebdding(abd, tnf).cpu().numpy())
                        barcodes.extend(data["bc"])
                embedding = np.concatenate(chunks, axis=0)
            np.savez(barcodes_path, barcodes)
            np.savez(latent_path, embedding)
        else:
            logging.info("latent and barcodes already saved")
        with open(os.path.join(model_path, "model_finished"), "w") as f:
            f.write("model finished")
