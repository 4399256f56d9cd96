"""Oracle-backed stand-in for freud_amd.engine.SaeEngine, for CPU tests of the HOST logic only
(train loop sequencing, checkpoints, loader order, data-parallel all-reduce).  Lives under tests/:
the product never imports it, and it is never the thing measured."""
import numpy as np
import torch

from oracle import sae_oracle as O


class OracleEngine:
    def __init__(self, variant, d_model, n_dict, max_rows, *, optimizer="radam", recon_alpha=1.0, k=0, auxk_alpha=0.0,
                 clip_thresh=1.0, weight_decay=0.0, device_id=0, multi_topk=False, autocast=True, **_):
        self.multi_topk = bool(multi_topk)
        self.autocast = bool(autocast)          # False: fp32 math (data-parallel exactness tests)
        self.dp_world, self.stats = 0, None
        self.variant, self.d, self.n, self.max_rows = variant, d_model, n_dict, max_rows
        self.optimizer, self.recon_alpha, self.k, self.auxk_alpha = optimizer, recon_alpha, k, auxk_alpha
        self.clip_thresh, self.weight_decay = clip_thresh, weight_decay
        self.st = O.OptState()
        self.P = {}
        self.dead_threshold = None
        self.nfsf = torch.zeros(n_dict, dtype=torch.long)
        nflat = sum(int(np.prod(s)) for s in self.param_shapes().values())
        self.flat = torch.zeros(nflat + 8 + (n_dict if variant == "topk" else 0))
        self._nflat = nflat
        self._latent = None

    def param_shapes(self):
        if self.variant == "l1":
            return {"decoder.weight": (self.d, self.n), "encoder_bias": (self.n,)}
        return {"encoder.weight": (self.n, self.d), "encoder.bias": (self.n,), "W_dec": (self.n, self.d), "b_dec": (self.d,)}

    def param_checksum(self):
        """Stand-in of sae_param_checksum: 64-bit words over the bytes of parameters / both moments + the step count, so that
        train()'s replica guard (freud_amd/dp.py: check_replicas) runs in the CPU data-parallel tests as well."""
        import hashlib

        def h(tensors):
            m = hashlib.blake2b(digest_size=8)
            for k in sorted(tensors):
                m.update(np.ascontiguousarray(tensors[k].numpy() if hasattr(tensors[k], "numpy") else tensors[k]).tobytes())
            return int.from_bytes(m.digest(), "little")
        return (h(self.P), h(self.st.exp_avg), h(self.st.exp_avg_sq), int(self.st.step))

    def get_topk_state(self):
        return self.nfsf.numpy().copy()

    def set_topk_state(self, a):
        self.nfsf = torch.as_tensor(a, dtype=torch.long).clone()

    def set_dead_feature_threshold(self, v):
        self.dead_threshold = v

    def set_topk_options(self, dead_feature_threshold, rows_per_file):
        self.dead_threshold = dead_feature_threshold

    def set_params(self, params):
        self.P = {k: torch.tensor(np.asarray(params[k], dtype=np.float32)).reshape(s).clone()
                  for k, s in self.param_shapes().items()}

    def get_params(self):
        return {k: v.numpy().copy() for k, v in self.P.items()}

    def set_opt_state(self, step, exp_avg, exp_avg_sq):
        self.st.step = int(step)
        self.st.exp_avg = {k: torch.tensor(np.asarray(exp_avg[k])).clone() for k in self.param_shapes()}
        self.st.exp_avg_sq = {k: torch.tensor(np.asarray(exp_avg_sq[k])).clone() for k in self.param_shapes()}

    def get_opt_state(self):
        z = {k: np.zeros(s, np.float32) for k, s in self.param_shapes().items()}
        m1 = {k: self.st.exp_avg[k].numpy().copy() if k in self.st.exp_avg else z[k] for k in z}
        m2 = {k: self.st.exp_avg_sq[k].numpy().copy() if k in self.st.exp_avg_sq else z[k] for k in z}
        return self.st.step, m1, m2

    def grad_tensor(self):
        return self.flat

    # data parallel: freud_amd.engine.SaeEngine.batch_stats / stats_tensor / set_dp_world
    def batch_stats(self, x, stream=None):
        x = x.detach().cpu()
        self.stats = O.l1_batch_stats(x) if self.variant == "l1" else O.topk_batch_stats(x.float())

    def stats_tensor(self):
        return self.stats

    def set_dp_world(self, world):
        self.dp_world = int(world)

    def _pack(self, grads, metrics):
        off = 0
        for k in self.param_shapes():
            g = grads[k].reshape(-1)
            self.flat[off:off + g.numel()] = g
            off += g.numel()
        self.flat[off:off + 8] = torch.tensor(metrics + [0.0] * (8 - len(metrics)))

    def forward_backward(self, x, stream=None):
        x = x.detach().cpu()
        if self.variant == "l1":
            W, b = self.P["decoder.weight"], self.P["encoder_bias"]
            W.copy_(O.normalize_columns(W))
            xf = x.reshape(-1, self.d).float()
            dp = {"dp_count": float(self.stats[0]), "dp_rows": float(self.stats[1])} if self.dp_world > 0 else {}
            f = O.l1_forward(xf, W, b, self.recon_alpha, self.autocast, **dp)
            dW, db = O.l1_backward(xf, W, b, f, self.recon_alpha, self.autocast)
            self._latent = f["c"]
            self._pack({"decoder.weight": dW, "encoder_bias": db},
                       [f["reconstruction_loss"].item(), f["l1_loss"].item(), f["mse"].item(), 0.0, float(f["local_count"])])
        else:
            P = self.P
            dead = self.nfsf > self.dead_threshold
            f = O.topk_forward(x.float(), P["encoder.weight"], P["encoder.bias"], P["W_dec"], P["b_dec"], self.k, dead,
                               self.auxk_alpha, self.autocast, self.multi_topk,
                               dp_stats=self.stats if self.dp_world > 0 else None)
            g = O.topk_backward(x.float(), P["encoder.weight"], P["encoder.bias"], P["W_dec"], P["b_dec"], f,
                                self.auxk_alpha, self.autocast)
            did = torch.zeros(self.n, dtype=torch.bool)
            did[f["fire_indices"].flatten()] = True
            self._did, self._rows = did, x.shape[0] * x.shape[1]
            self._latent = f["dense"].float()
            self._pack({"encoder.weight": g["W_enc"], "encoder.bias": g["b_enc"], "W_dec": g["W_dec"], "b_dec": g["b_dec"]},
                       [f["fvu"].item(), f["auxk_loss"].item(), f["mse"].item(), 0.0, 0.0,
                        float(dead.float().mean()) / max(self.dp_world, 1), f["multi_topk_fvu"].item()])
            self.flat[-self.n:] = did.float()        # did_fire flags ride in the all-reduced buffer (OR == sum > 0)

    def optimizer_step(self, lr, grad_scale=1.0, stream=None):
        off, grads = 0, {}
        for k, s in self.param_shapes().items():
            n = int(np.prod(s))
            grads[k] = (self.flat[off:off + n] * grad_scale).reshape(s).clone()
            off += n
        order = ["encoder_bias", "decoder.weight"] if self.variant == "l1" else ["W_dec", "b_dec", "encoder.weight", "encoder.bias"]
        gn, clipped = O.clip_grad_norm([grads[k] for k in order], self.clip_thresh)
        cg = dict(zip(order, clipped))
        params = {k: self.P[k] for k in order}
        if self.optimizer == "radam":
            O.radam_step(params, cg, self.st, lr, eps=1e-5, weight_decay=self.weight_decay)
        else:
            O.adam_step(params, cg, self.st, lr)
        self.flat[off + 3] = gn
        if grad_scale != 1.0:
            self.flat[off:off + 3] *= grad_scale
            self.flat[off + 5] *= grad_scale
        if self.variant == "topk":               # train_sae.py:443-446 with the (possibly summed) did_fire flags
            rows = int(self.stats[0]) if self.dp_world > 0 else self._rows * (round(1.0 / grad_scale) if grad_scale > 0 else 1)
            self.nfsf += rows
            self.nfsf[self.flat[-self.n:] > 0] = 0

    def step(self, x, lr, stream=None):
        self.forward_backward(x)
        self.optimizer_step(lr, 1.0)

    def eval(self, x, stream=None):
        x = x.detach().cpu()
        W, b = self.P["decoder.weight"], self.P["encoder_bias"]
        W.copy_(O.normalize_columns(W))
        f = O.l1_forward(x.reshape(-1, self.d).float(), W, b, self.recon_alpha, True)
        self._latent = f["c"]
        off = self._nflat
        self.flat[off:off + 5] = torch.tensor([f["reconstruction_loss"].item(), f["l1_loss"].item(), f["mse"].item(), 0.0,
                                               float(f["count"])])

    def metrics(self, stream=None):
        return self.flat[self._nflat:self._nflat + 8].numpy().copy()

    def latent_colmax(self, stream=None):
        return self._latent.abs().reshape(-1, self.n).max(0).values.numpy()

    def close(self):
        pass
