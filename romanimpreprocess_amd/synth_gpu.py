"""The synthetic CALDIR and Level-1 ramp of ``synth.py`` generated on the GPU with torch (f64): a non-periodic full
4096 x 4096 frame in seconds where the numpy generator needs minutes of host time (inverse linearity by Newton steps on
16.7 M pixels per group).  Test and benchmark INPUT generation only -- nothing here is on the calibration path; the
recipe (sky + 25 Gaussian sources through IPC and the inverted linearity curve, read noise,
row-correlated noise seen by the reference pixels and the reference output, cosmic-ray steps, saturation) and the
array layout are those of ``synth.make_caldir`` / ``synth.make_ramp``; the random streams are torch's, so the data
differ from the numpy generator's for the same seed.  Results come back as numpy arrays (same dict layout).
"""

import numpy as np
import torch

from . import pars, synth
from .dqflags import pixel

F64 = torch.float64


def _gen(dev, seed):
    g = torch.Generator(device=dev)
    g.manual_seed(int(seed))
    return g


def _legendre_tail(z, coefs, l0):
    pm, p = torch.ones_like(z), z.clone()
    dpm, dp = torch.zeros_like(z), torch.ones_like(z)
    val, der = torch.zeros_like(z), torch.zeros_like(z)
    L, top = 1, l0 + coefs.shape[0] - 1
    while L <= top:
        if L >= l0:
            val += coefs[L - l0] * p
            der += coefs[L - l0] * dp
        pn = ((2 * L + 1) * z * p - L * pm) / (L + 1)
        dpn = ((2 * L + 1) * (p + z * dp) - L * dpm) / (L + 1)
        pm, p, dpm, dp = p, pn, dp, dpn
        L += 1
    return val, der


def make_caldir(ny=pars.nside, nx=pars.nside, read_pattern=None, frame_time=synth.FRAME_TIME, p_order=8, seed=1000,
                gain_dtype=np.float32, ipc_dtype=np.float32, nb=pars.nborder, bias_amplitude=0.0, bad_lin_frac=0.0,
                high_order_scale=0.05, device=0):
    dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    rp = synth.READ_PATTERN_8 if read_pattern is None else read_pattern
    g = _gen(dev, seed)
    G = len(rp)
    t = torch.tensor(synth.group_times(rp, frame_time), dtype=F64, device=dev)
    y = torch.arange(ny, dtype=F64, device=dev)[:, None].expand(ny, nx)
    x = torch.arange(nx, dtype=F64, device=dev)[None, :].expand(ny, nx)
    nya, nxa = ny - 2 * nb, nx - 2 * nb

    def normal(*shape):
        return torch.randn(*shape, dtype=F64, device=dev, generator=g)

    def uniform(*shape):
        return torch.rand(*shape, dtype=F64, device=dev, generator=g)

    def host(a, dtype):
        return a.to(dtype).cpu().numpy()

    cal = {}
    dark_slope = 0.005 * 10.0 ** normal(ny, nx)
    dark_slope[:nb] = 0
    dark_slope[-nb:] = 0
    dark_slope[:, :nb] = 0
    dark_slope[:, -nb:] = 0
    bias = 13000 + 200 * torch.cos(2.0 * np.pi * x / 256.0) + 100 * torch.sin(2.0 * np.pi * y / 256.0) ** 3
    cal["dark"] = {
        "data": host(torch.clamp((bias[None] + dark_slope[None] * t[:, None, None]).to(torch.float32), 0.0, 65535.0), torch.float32),
        "dq": np.zeros((ny, nx), dtype=np.uint32),
        "dark_slope": host(dark_slope, torch.float32),
        "dark_slope_err": np.zeros((ny, nx), dtype=np.float32),
    }
    tdt = {np.float32: torch.float32, np.float64: torch.float64}
    cal["gain"] = {"data": host(torch.clamp(1.5 + 0.03 * normal(ny, nx), 1.4, 1.6), tdt[gain_dtype]),
                   "dq": np.zeros((ny, nx), dtype=np.uint32)}

    kd = tdt[ipc_dtype]
    K = torch.zeros((3, 3, nya, nxa), dtype=kd, device=dev)
    K[0, 1] = K[2, 1] = 0.015
    K[1, 0] = K[1, 2] = 0.013
    K[0, 0] = K[2, 2] = K[0, 2] = K[2, 0] = 0.002
    K *= (1.0 + 0.05 * normal(1, 1, nya, nxa)).to(kd)
    K[0, :, 0, :] = 0.0
    K[:, 0, :, 0] = 0.0
    K[-1, :, -1, :] = 0.0
    K[:, -1, :, -1] = 0.0
    K[1, 1] = 0.0
    K[1, 1] = 1.0 - K.sum(dim=(0, 1))
    cal["ipc4d"] = {"data": K.cpu().numpy(), "dq": np.zeros((ny, nx), dtype=np.uint32)}
    del K

    Smin = torch.clamp(5000 + 500 * torch.cos((x + 3 * y) / 100.0), 0.5, 65534.5).to(torch.float32)
    Smax = torch.clamp(56000 + 10000 * uniform(ny, nx), 0.5, 65534.5).to(torch.float32)
    Sref = (Smin + 300 + 100 * (x % 2)).to(torch.float32)
    coefs = torch.zeros((p_order + 1, ny, nx), dtype=torch.float32, device=dev)
    coefs[2] = (20 + 180 * uniform(ny, nx)).to(torch.float32)
    for L in range(3, p_order + 1):
        coefs[L] = ((high_order_scale * 40.0 / L**2) * normal(ny, nx)).to(torch.float32)
    z = 2 * (Sref.to(F64) - Smin) / (Smax.to(F64) - Smin) - 1
    val, der = _legendre_tail(z, coefs[2:].to(F64), 2)
    c1 = (Smax.to(F64) - Smin) / 2.0 - der
    coefs[1] = c1.to(torch.float32)
    coefs[0] = (-(c1 * z) - val).to(torch.float32)
    lin_dq = np.zeros((ny, nx), dtype=np.uint32)
    if bad_lin_frac > 0:
        lin_dq |= np.where((uniform(ny, nx) < bad_lin_frac).cpu().numpy(), pixel.NO_LIN_CORR, 0).astype(np.uint32)
    cal["linearitylegendre"] = {"data": coefs.cpu().numpy(), "dq": lin_dq, "Smin": Smin.cpu().numpy(), "Smax": Smax.cpu().numpy(),
                                "Sref": Sref.cpu().numpy()}

    mask = np.zeros((ny, nx), dtype=np.uint32)
    mask[:nb, :] |= pixel.REFERENCE_PIXEL
    mask[-nb:, :] |= pixel.REFERENCE_PIXEL
    mask[:, :nb] |= pixel.REFERENCE_PIXEL
    mask[:, -nb:] |= pixel.REFERENCE_PIXEL
    ds = cal["dark"]["dark_slope"]
    mask |= np.where(ds > 0.25, np.where(ds > 12.5, pixel.HOT, pixel.WARM), 0).astype(np.uint32)
    cal["mask"] = {"dq": mask}

    pflat = (0.95 + 0.1 * (x / nx - 1) - 0.2 * (y / ny * (1 - y / ny))).to(torch.float32)
    pflat[:nb] = 0
    pflat[-nb:] = 0
    pflat[:, :nb] = 0
    pflat[:, -nb:] = 0
    cal["flat"] = {"data": pflat.cpu().numpy(), "dq": np.zeros((ny, nx), dtype=np.uint32)}

    med = np.full((ny, pars.channelwidth), 29000.0, dtype=np.float32)
    std = np.full((ny, pars.channelwidth), 4.0, dtype=np.float32)
    for r in range(0, ny, 256):
        std[r] = 5
        med[r] += 30
        if r + 1 < ny:
            med[r + 1] += 15
    cal["read"] = {
        "anc": {"U_PINK": 0.4, "C_PINK": 0.8},
        "data": host(6.0 + 5.0 * uniform(ny, nx), torch.float32),
        "resetnoise": host(25.0 + 5.0 * uniform(ny, nx), torch.float32),
        "amp33": {"valid": True, "med": med, "std": std, "M_PINK": 0.8, "RU_PINK": 1.0},
    }
    cal["saturation"] = {"data": host(torch.clamp(Smax - 50, min=1.5), torch.float32), "dq": np.zeros((ny, nx), dtype=np.uint32)}
    bc = torch.zeros((G, nya, nxa), dtype=torch.float32, device=dev)
    if bias_amplitude:
        bc += (bias_amplitude * normal(G, nya, nxa)).to(torch.float32)
    cal["biascorr"] = {"data": bc.cpu().numpy(), "t0": float(t[1])}
    return cal


def _ipc_forward(img, K):
    ny, nx = img.shape
    out = img * K[1, 1]
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            if dy == 0 and dx == 0:
                continue
            ys, xs = slice(max(0, -dy), ny - max(0, dy)), slice(max(0, -dx), nx - max(0, dx))
            yd, xd = slice(max(0, dy), ny - max(0, -dy)), slice(max(0, dx), nx - max(0, -dx))
            out[yd, xd] += img[ys, xs] * K[1 + dy, 1 + dx, ys, xs]
    return out


class RampFactory:
    """Seeded Level-1 ramps of one scene / calibration set, generated on the device and LEFT there (the many-realisations
    harness makes hundreds of them: the calibration arrays are uploaded once, each ramp takes about half a second of f64
    torch work for a full frame and never crosses PCIe)."""

    def __init__(self, cal, read_pattern=None, frame_time=synth.FRAME_TIME, nb=pars.nborder, device=0, exclude_first=True,
                 saturation_backup=1):
        self.dev = dev = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
        self.rp = synth.READ_PATTERN_8 if read_pattern is None else read_pattern
        self.G, self.nb, self.frame_time = len(self.rp), nb, frame_time
        self.exclude_first, self.backup = exclude_first, saturation_backup
        lin = cal["linearitylegendre"]
        self.ny, self.nx = lin["Smin"].shape
        self.t = synth.group_times(self.rp, frame_time)
        self.nread = [float(len(r)) for r in self.rp]

        def dv(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(dev).to(F64)

        self.gain, self.dark_slope = dv(cal["gain"]["data"]), dv(cal["dark"]["dark_slope"])
        self.sigma_read, self.sat_level = dv(cal["read"]["data"]), dv(cal["saturation"]["data"])
        self.K = dv(cal["ipc4d"]["data"])
        self.Smin = dv(lin["Smin"])
        self.span = dv(lin["Smax"]) - self.Smin
        self.coef, self.Sref = dv(lin["data"]), dv(lin["Sref"])
        self.dark = torch.from_numpy(np.ascontiguousarray(cal["dark"]["data"][:self.G])).to(dev)  # f32, widened per group
        b = cal["biascorr"]["data"] if "biascorr" in cal else None
        self.bias = None if b is None else torch.from_numpy(np.ascontiguousarray(b[b.shape[0] - self.G:])).to(dev)
        a33 = cal["read"]["amp33"]
        self.a33_med, self.a33_std, self.m_pink = dv(a33["med"]), dv(a33["std"]), float(a33["M_PINK"])
        self.border = torch.ones((self.ny, self.nx), dtype=torch.bool, device=dev)
        self.border[nb:self.ny - nb, nb:self.nx - nb] = False
        self.mask = torch.from_numpy(np.array(cal["mask"]["dq"], dtype=np.uint32).view(np.int32)).to(dev)

    def _phi(self, S):
        z = 2 * (S - self.Smin) / self.span - 1
        val, der = _legendre_tail(z, self.coef[1:], 1)
        return self.coef[0] + val, der * 2.0 / self.span

    def make(self, seed, rate, cr_frac=1e-3, poisson=False):
        """``rate``: (ny, nx) tensor or array of linearised DN/s.  ``poisson``: accumulate Poisson electron counts read by read
        (rate * gain * frame_time per read, averaged over the reads of each group -- the covariance structure the ramp-fit
        weights and error estimates assume) instead of the noiseless signal.  Returns device tensors: data (G,ny,nx) int16 (the u16 bits),
        amp33 (G,ny,128) int16, groupdq (G,ny,nx) uint8 (after dq-init + saturation flagging, DO_NOT_USE on group 0 when the
        first group is excluded), pixeldq (ny,nx) int32."""
        dev, G, ny, nx, nb = self.dev, self.G, self.ny, self.nx, self.nb
        g = _gen(dev, 7_000_003 + int(seed))
        zero = torch.zeros((), dtype=F64, device=dev)
        one = torch.ones((), dtype=F64, device=dev)

        def normal(*shape):
            return torch.randn(*shape, dtype=F64, device=dev, generator=g)

        def uniform(*shape):
            return torch.rand(*shape, dtype=F64, device=dev, generator=g)

        rate = rate if torch.is_tensor(rate) else torch.from_numpy(np.ascontiguousarray(rate)).to(dev)
        act = (slice(nb, ny - nb), slice(nb, nx - nb))
        total_rate = rate.to(F64) + self.dark_slope
        cr_mask = uniform(ny, nx) < cr_frac
        cr_grp = torch.randint(2, max(G, 3), (ny, nx), device=dev, generator=g)
        cr_amp = torch.where(cr_mask, 50.0 * 100.0 ** uniform(ny, nx), zero)
        data = torch.empty((G, ny, nx), dtype=torch.int16, device=dev)
        amp33 = torch.empty((G, ny, pars.channelwidth), dtype=torch.int16, device=dev)
        groupdq = torch.zeros((G, ny, nx), dtype=torch.uint8, device=dev)
        S_guess = self.Sref
        sat = int(pixel.SATURATED)
        sig_groups = None
        if poisson:
            lam = torch.clamp(total_rate * self.gain * self.frame_time, min=0.0)   # electrons per read
            q = torch.zeros_like(lam)
            sig_groups, last = [], -1
            for gi, reads in enumerate(self.rp):
                acc = torch.zeros_like(lam)
                for r in reads:
                    while last < r:     # electrons collected up to the END of read r (read 0 ends one frame after the reset)
                        q = q + torch.poisson(lam, generator=g)
                        last += 1
                    acc += q
                # the deterministic model samples the signal at the mean read index: keep its time origin
                sig_groups.append(acc / len(reads) / self.gain - total_rate * self.frame_time)
        for gi in range(G):
            base = sig_groups[gi] if poisson else total_rate * float(self.t[gi])
            sig = base + torch.where(cr_grp <= gi, cr_amp, zero)
            conv = sig.clone()
            conv[act] = _ipc_forward(sig[act] * self.gain[act], self.K) / self.gain[act]
            S = S_guess.clone()
            for _ in range(4):
                val, der = self._phi(S)
                S = S - (val - conv) / torch.where(der.abs() > 0.2, der, one)
            S_guess = S
            row_noise = 3.0 * normal(ny, 1)
            raw = S + row_noise + self.sigma_read / np.sqrt(self.nread[gi]) * normal(ny, nx)
            if self.bias is not None:
                raw[act] += self.bias[gi].to(F64)
            ref = self.dark[gi].to(F64) + row_noise + self.sigma_read * normal(ny, nx)
            raw = torch.where(self.border, ref, raw)
            sat_now = (raw >= self.sat_level) & ~self.border
            raw = torch.where(sat_now, torch.minimum(raw, self.sat_level + 200.0), raw)
            data[gi] = torch.clamp(torch.round(raw), 0, 65535).to(torch.int32).to(torch.int16)  # wraps to the u16 bit pattern
            groupdq[gi] |= torch.where(sat_now, sat, 0).to(torch.uint8)
            r33 = self.a33_med + self.m_pink * row_noise + self.a33_std * normal(ny, pars.channelwidth)
            amp33[gi] = torch.clamp(torch.round(r33), 0, 65535).to(torch.int32).to(torch.int16)
        # saturation is sticky forward in time and flagged `backup` groups early; group 0 is not checked
        for gi in range(1, G):
            groupdq[gi] |= groupdq[gi - 1] & sat
        for _ in range(self.backup):
            for gi in range(1, G - 1):
                groupdq[gi] |= groupdq[gi + 1] & sat
        groupdq[0] &= 0xFF ^ sat
        if self.exclude_first:
            groupdq[0] |= int(pixel.DO_NOT_USE)
        return data, amp33, groupdq, self.mask


def make_ramp(cal, read_pattern=None, frame_time=synth.FRAME_TIME, seed=1, cr_frac=1e-3, rate=None, exclude_first=True,
              nb=pars.nborder, saturation_backup=1, device=0):
    """``synth.make_ramp`` on the device; numpy arrays back (same dict layout)."""
    f = RampFactory(cal, read_pattern, frame_time, nb, device, exclude_first, saturation_backup)
    if rate is None:
        rate = synth.make_rate_image(f.ny, f.nx, seed, nb=nb)
    data, amp33, gq, _mask = f.make(seed, np.asarray(rate, dtype=np.float64), cr_frac)
    return {"data": data.cpu().numpy().view(np.uint16), "amp33": amp33.cpu().numpy().view(np.uint16), "groupdq": gq.cpu().numpy(),
            "pixeldq": np.array(cal["mask"]["dq"], dtype=np.uint32, copy=True), "read_pattern": f.rp, "frame_time": frame_time,
            "rate": np.asarray(rate, dtype=np.float32)}
