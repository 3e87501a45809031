"""Intra-frame sharding: ONE image split into row bands over the GPUs of a node.

The frame-parallel path (frames.py) needs no data-path collective; this one does -- it is the path's only real exchange
step.  Rank r owns rows [r0, r1) of an R x C image and holds them plus HALO = p//2 + 1 rows (2 for p = 3) of real image data on every side
that is not an image border (k_detect reads x that many rows away from the pixel it scores).  Between the sweeps the ranks
all-reduce a handful of doubles (include/wm.h, wm_band_*):

    embed : 44 Gram sums (SUM) -> solve -> {max|e| (MAX), sum (|e| W)^2 (SUM)} -> strength -> embed the owned rows
    detect: halo rows of y from the neighbour bands (point to point) -> 44 Gram sums (SUM) -> solve
            -> {<e_u,e_w>, |e_u|^2, |e_w|^2} (SUM) -> corr

`torch.distributed` carries the exchange: backend "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the tests.  The all-reduces
are 8..352 bytes: latency-bound, a few tens of microseconds each against sweeps of 1/G of the image.

Two forms of the same protocol:
  * device-resident (backend "nccl"): the totals never leave HBM.  The engine's slot runs on torch's current stream, every
    phase only enqueues (wm.h wm_band_*_dev), and the collectives are RCCL calls on device tensors ordered on that stream:
    one all-reduce of 44 doubles, one all-gather of {max, sum} pairs (folded in rank order on the device), one all-reduce
    of 3 doubles; halo rows by ncclSend / ncclRecv (batch_isend_irecv on device tensors).  The host reads one float at the end.
  * host exchange (backend "gloo", the CPU tests): the synchronous wm_band_* calls hand the totals to the host, small CPU
    tensors are all-reduced."""
import importlib

import numpy as np
import torch
import torch.distributed as dist

HALO = 2  # p = 3


def halo_rows(p):
    """halo rows a band needs on every interior side: k_detect scores e_u = u - c.nbrs(u), so it reads the mask one row
    away from the pixel it scores, and the mask reads x another p//2 rows away (ME: 1)"""
    return max(HALO, p // 2 + 1)


def band_rows(rows, rank, world):
    """owned rows [r0, r1) of rank `rank`: contiguous bands of near-equal height"""
    base, rem = divmod(rows, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def band_with_halo(rows, rank, world, halo=HALO):
    """(g0, g1, own_lo, own_hi): rows [g0, g1) the rank holds, and the owned rows in that band's coordinates"""
    r0, r1 = band_rows(rows, rank, world)
    g0, g1 = max(0, r0 - halo), min(rows, r1 + halo)
    return g0, g1, r0 - g0, r1 - g0


def _allreduce(values, op, device):
    t = torch.tensor(values, dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=op)
    return t.cpu().numpy()


class BandedWatermark:
    """One rank's part of a watermark engine over a row-sharded image.

    rows, cols: the WHOLE image; W: the whole [rows, cols] watermark (numpy) or this rank's band of it with halo rows.
    coll_device: where the tiny collective tensors live ("cuda" for nccl, "cpu" for gloo)."""

    def __init__(self, rows, cols, W, p, psnr, rank, world, device=0, coll_device=None):
        wm = importlib.import_module(__package__)
        self.wm = wm
        self.rows, self.cols, self.rank, self.world = rows, cols, rank, world
        self.halo = halo_rows(p)
        self.g0, self.g1, self.own_lo, self.own_hi = band_with_halo(rows, rank, world, self.halo)
        Wb = np.ascontiguousarray(W[self.g0:self.g1] if W.shape[0] == rows else W, dtype=np.float32)
        assert Wb.shape == (self.g1 - self.g0, cols)
        self.eng = wm.Watermark(self.g1 - self.g0, cols, Wb, p, psnr, device=device)
        self.eng.band_configure(self.own_lo, self.own_hi, rows)
        self.coll_device = coll_device or ("cuda" if dist.is_initialized() and dist.get_backend() == "nccl" else "cpu")
        # device-resident exchange when the collectives run on the GPU; force_collective issues them even with one rank (the
        # one-GPU rehearsal of the RCCL path)
        self.on_device = str(self.coll_device).startswith("cuda")
        self.force_collective = False
        if self.on_device:
            dev = torch.device("cuda", device)
            self._tot = torch.zeros(44, dtype=torch.float64, device=dev)
            self._ms = torch.zeros(2, dtype=torch.float64, device=dev)
            self._parts = torch.zeros(2 * world, dtype=torch.float64, device=dev)
            self._sums = torch.zeros(3, dtype=torch.float64, device=dev)
            self._a = torch.zeros(1, dtype=torch.float32, device=dev)
            self._corr = torch.zeros(1, dtype=torch.float32, device=dev)

    def close(self):
        self.eng.close()

    def local_view(self, full):
        """this rank's band (owned rows + halo) of a whole-image tensor [rows, cols]"""
        return full[self.g0:self.g1].contiguous()

    def _solve(self, band):
        tot = _allreduce(self.eng.gram_totals(band), dist.ReduceOp.SUM, self.coll_device)
        return self.eng.band_solve(tot)

    # ---- device-resident form -------------------------------------------------------------------------------------------
    def _collective(self):
        return dist.is_initialized() and (self.world > 1 or self.force_collective)

    def _solve_dev(self, band):
        self.eng.band_gram_dev(band, self._tot)
        if self._collective():
            dist.all_reduce(self._tot)          # RCCL, 352 bytes, ordered on the current stream
        self.eng.band_solve_dev(self._tot)

    def _embed_dev(self, band, mask):
        ME = int(self.wm.MASK_TYPE.ME)
        self.eng.set_stream_current()
        if int(mask) == ME:
            self._solve_dev(band)
        self.eng.band_stats_dev(band, mask, self._ms)
        if self._collective():
            dist.all_gather_into_tensor(self._parts, self._ms)   # {max|e|, sum} of every band; folded in rank order on the device
            nparts = self.world
        else:
            self._parts[:2] = self._ms
            nparts = 1
        out = band.clone()
        self.eng.band_embed_dev(band, band, out, mask, self._parts, nparts, self._a)
        a = float(self._a.item())               # the one host read of the operation
        return out, (None if a != a else a)     # NaN: unsolvable on every rank alike, out is the input (Watermark.cpp:164-165)

    def _detect_dev(self, band, mask):
        self.eng.set_stream_current()
        self._solve_dev(band)
        self.eng.band_detect_sums_dev(band, mask, self._sums)
        if self._collective():
            dist.all_reduce(self._sums)
        self.eng.band_corr_dev(self._sums, self._corr)
        return float(self._corr.item())

    def embed(self, band, mask):
        """band: [g1-g0, cols] device tensor (owned rows + halo).  Returns (y_band, strength or None): y_band holds the
        watermarked OWNED rows; its halo rows are copies of the input (exchange_halos refreshes them for detect)."""
        if self.on_device:
            return self._embed_dev(band, mask)
        ME = int(self.wm.MASK_TYPE.ME)
        if int(mask) == ME and self._solve(band) != 0:
            return band.clone(), None  # unsolvable on every rank alike: passthrough (Watermark.cpp:164-165)
        mx, ss = self.eng.band_stats(band, mask)
        mx = float(_allreduce([mx], dist.ReduceOp.MAX, self.coll_device)[0])
        ss = float(_allreduce([ss], dist.ReduceOp.SUM, self.coll_device)[0])
        out = band.clone()
        a = self.eng.band_embed(band, band, out, mask, mx, ss)
        return out, a

    def exchange_halos(self, band):
        """refresh the halo rows of `band` with the neighbours' owned rows (after embed wrote the owned rows only)"""
        if self.world == 1 or not dist.is_initialized():
            return band
        cpu = self.coll_device == "cpu"
        HALO = self.halo
        ops, recv = [], []
        up, dn = self.rank - 1, self.rank + 1
        def stage(t):
            return t.cpu().contiguous() if cpu else t.contiguous()
        if up >= 0:
            send = stage(band[self.own_lo:self.own_lo + HALO])
            buf = torch.empty_like(send)
            ops += [dist.P2POp(dist.isend, send, up), dist.P2POp(dist.irecv, buf, up)]
            recv.append((buf, slice(self.own_lo - HALO, self.own_lo)))
        if dn < self.world:
            send = stage(band[self.own_hi - HALO:self.own_hi])
            buf = torch.empty_like(send)
            ops += [dist.P2POp(dist.isend, send, dn), dist.P2POp(dist.irecv, buf, dn)]
            recv.append((buf, slice(self.own_hi, self.own_hi + HALO)))
        for w in dist.batch_isend_irecv(ops):
            w.wait()
        for buf, sl in recv:
            band[sl] = buf.to(band.device)
        return band

    def detect(self, band, mask):
        """correlation of the whole image, identical on every rank; 0.0 for an unsolvable system (Watermark.cpp:246-247)"""
        if self.on_device:
            return self._detect_dev(band, mask)
        if self._solve(band) != 0:
            return 0.0
        d, nu, nw = self.eng.band_detect_sums(band, mask)
        d, nu, nw = _allreduce([d, nu, nw], dist.ReduceOp.SUM, self.coll_device)
        return float(np.float32(d) / np.float32(np.sqrt(nw) * np.sqrt(nu)))
