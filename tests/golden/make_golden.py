"""Generates the committed golden fixtures from the ORACLE (this build's PyTorch-CPU restatement of
arch_spec).  There is no Python reference to import — /root/reference holds README.md and
requirements.txt only — so these vectors pin the oracle against silent drift between torch builds /
hosts, not against the reference ("parity unpinned").

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz

Inputs are regenerated from seeds by s3r.synthetic_pairs / s3r.seed_module, so only outputs are stored.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

import s3r  # noqa: E402
from oracle import s2v_oracle as O  # noqa: E402


def main():
    torch.manual_seed(0)
    # --- Stereo2Voxel, B=2, weights seed 0, inputs seed 0 (BASELINE.json configs[0] shape)
    m = O.OracleStereo2Voxel().eval()
    s3r.seed_module(m, 0)
    left, right = s3r.synthetic_pairs(2, seed=0)
    with torch.no_grad():
        feats = m.encoder(torch.cat([left, right]))
        vol = O.cost_volume(feats[:2], feats[2:])
        occ = m.decoder(vol)
        latent = m.decoder(vol, upto="v6")
    np.savez_compressed(os.path.join(HERE, "s2v_b2_seed0.npz"),
                        features=feats.numpy(), occupancy=occ.numpy(),
                        volume_sum=np.float64(vol.double().sum().item()),
                        volume_abs_sum=np.float64(vol.double().abs().sum().item()),
                        latent_sum=np.float64(latent.double().sum().item()),
                        latent_abs_sum=np.float64(latent.double().abs().sum().item()))
    # --- Stereo2Point, B=2, weights seed 1, inputs seed 2
    mp = O.OracleStereo2Point().eval()
    s3r.seed_module(mp, 1)
    l2, r2 = s3r.synthetic_pairs(2, seed=2)
    with torch.no_grad():
        pts = mp(l2, r2)
    # --- Chamfer known-answer: hand-checkable clouds + a seeded random pair
    p = torch.tensor([[[0.0, 0, 0], [1, 0, 0], [0, 2, 0]]])
    q = torch.tensor([[[0.0, 0, 1], [3, 0, 0]]])
    d1, d2, i1, i2 = O.chamfer_distance(p, q)
    g = torch.Generator().manual_seed(4)
    pr, qr = torch.rand(2, 256, 3, generator=g), torch.rand(2, 300, 3, generator=g)
    rd1, rd2, ri1, ri2 = O.chamfer_distance(pr, qr)
    np.savez_compressed(os.path.join(HERE, "s2p_chamfer.npz"),
                        points=pts.numpy(),
                        kat_p=p.numpy(), kat_q=q.numpy(), kat_d1=d1.numpy(), kat_d2=d2.numpy(),
                        kat_i1=i1.numpy(), kat_i2=i2.numpy(),
                        rnd_d1=rd1.numpy(), rnd_d2=rd2.numpy(), rnd_i1=ri1.numpy(), rnd_i2=ri2.numpy())
    print("wrote", os.listdir(HERE))


if __name__ == "__main__":
    main()
