#!/usr/bin/env python3
"""Summarise a BODAHIP_CBIG_TSTAMP file (clock stamps of conv_big_f32.hip launches built with -DTSTAMP=1): cbig_tl_parse.py file [n_last]"""
import sys
import numpy as np


def analyze(path, n_last=1):
    launches = []; cur = None
    for line in open(path):
        if line.startswith("launch"): cur = []; launches.append((line.strip(), cur))
        else: cur.append([int(x) for x in line.split()])
    for hdr, L in launches[-n_last:]:
        T = np.array(L, dtype=np.float64); t0 = T[:, 0].min()
        pro, loop, iss = (T[:, 2] - T[:, 0]) / 100.0, (T[:, 3] - T[:, 2]) / 100.0, (T[:, 4] - T[:, 3]) / 100.0
        end = (T[:, 5] - t0) / 100.0; cyc = T[:, 7] - T[:, 6]
        bw = np.array([int(r[1]) >> 4 for r in L], dtype=np.float64)
        print(f"{hdr}: {len(L)} wgs, end {end.max():.1f} us | prologue {np.median(pro):.1f} | K loop {np.median(loop):.1f} us = {np.median(cyc):.0f} cyc @ {np.median(cyc / np.maximum(loop, 1e-9)) / 1e3:.3f} GHz | "
              f"store issue {np.median(iss):.1f} | wave0 at barriers {100 * np.median(bw / np.maximum(cyc, 1)):.1f} % | stager: filt {np.median(T[:, 8]):.0f} pel {np.median(T[:, 9]):.0f} lds {np.median(T[:, 10]):.0f} barrier {np.median(T[:, 11]):.0f}")


if __name__ == "__main__":
    analyze(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 1)
