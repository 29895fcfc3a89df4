"""Device-side twin of tools/simulate_reads.py (same model, same distributions, torch RNG) so that bench.py can
synthesise tens of thousands of 32 k-sample reads directly in HBM in seconds."""
import numpy as np
import torch

from tools.simulate_reads import CAL_DIGITISATION, CAL_OFFSET, CAL_RANGE, load_template_model


def simulate_reads_torch(codes, contig_lens, n_reads, seed=42, device="cuda", read_bases=3600, off_target=0.10,
                         dwell_mean=8.9, noise_sd=1.5, chunk=1024):
    """-> dict(signal int16 tensor on `device`, offsets uint64 ndarray[n+1], contig/pos/strand ndarrays)"""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    means, stdvs = load_template_model()
    means_t, stdvs_t = torch.from_numpy(means).to(dev), torch.from_numpy(stdvs).to(dev)
    codes_t = torch.as_tensor(codes, dtype=torch.uint8, device=dev)
    lens = torch.as_tensor(np.asarray(contig_lens, dtype=np.int64), device=dev)
    contig_off = torch.cat((torch.zeros(1, dtype=torch.int64, device=dev), torch.cumsum(lens, 0)))
    usable = torch.clamp(lens - read_bases, min=0)
    pw = usable.double() / usable.sum().double()
    nk = read_bases - 4
    ar = torch.arange(read_bases, device=dev)
    parts, rls, cgs, pss, sts = [], [], [], [], []
    for r0 in range(0, n_reads, chunk):
        R = min(chunk, n_reads - r0)
        off = torch.rand(R, generator=g, device=dev) < off_target
        cg = torch.multinomial(pw, R, replacement=True, generator=g)
        ps = (torch.rand(R, generator=g, device=dev, dtype=torch.float64) * (usable[cg] + 1).double()).long()
        st = torch.randint(0, 2, (R,), generator=g, device=dev)
        gs = codes_t[(contig_off[cg] + ps)[:, None] + ar[None, :]]
        rc = 3 - gs.flip(1)
        seqs = torch.where((st == 0)[:, None], gs, rc)
        rnd = torch.randint(0, 4, (R, read_bases), generator=g, device=dev, dtype=torch.uint8)
        seqs = torch.where(off[:, None], rnd, seqs).long()
        k = seqs[:, 0:nk]
        for j in range(1, 5):
            k = (k << 2) | seqs[:, j:j + nk]
        lv = means_t[k] + stdvs_t[k] * torch.randn(R, nk, generator=g, device=dev)
        scale = 1.0 + 0.05 * torch.randn(R, generator=g, device=dev)
        shift = 5.0 * torch.randn(R, generator=g, device=dev)
        lv = lv * scale[:, None] + shift[:, None]
        dwell = torch.empty(R, nk, device=dev).geometric_(1.0 / dwell_mean, generator=g).long()
        rl = dwell.sum(1)
        pa = torch.repeat_interleave(lv.flatten(), dwell.flatten())
        pa = pa + noise_sd * torch.randn(pa.numel(), generator=g, device=dev)
        raw = torch.round(pa * (CAL_DIGITISATION / CAL_RANGE)) - CAL_OFFSET
        parts.append(raw.clamp_(0, 32767).to(torch.int16))
        rls.append(rl.cpu())
        cgs.append(torch.where(off, torch.full_like(cg, -1), cg).cpu())
        pss.append(torch.where(off, torch.zeros_like(ps), ps).cpu())
        sts.append(torch.where(off, torch.zeros_like(st), st).cpu())
    signal = torch.cat(parts)
    lens_all = torch.cat(rls).numpy().astype(np.uint64)
    offsets = np.concatenate(([0], np.cumsum(lens_all))).astype(np.uint64)
    return dict(signal=signal, offsets=offsets, contig=torch.cat(cgs).numpy().astype(np.int32),
                pos=torch.cat(pss).numpy(), strand=torch.cat(sts).numpy().astype(np.int8))
