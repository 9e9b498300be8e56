"""ORACLE helper — TEST INFRASTRUCTURE ONLY.  Deterministic, platform-independent parameter fill
(numpy's legacy MT19937 RandomState) so the fixture generator (which runs the REFERENCE modules in
the build container) and the tests (which run the oracle and the HIP path) use bit-identical
weights without committing multi-megabyte weight files."""
import zlib

import numpy as np
import torch


def fill_module(module, seed=1234, gain=1.0):
    """Overwrite every parameter / float buffer of `module` in place, keyed by state_dict name.
    Conv/linear weights ~ N(0, sqrt(2/fan_in)) (keeps 20-layer stacks O(1)), biases ~ 0.05*N,
    norm weights ~ 1 + 0.1*N (`gain` scales conv/linear weights: residual nets need < 1), PReLU slopes ~ 0.25 + 0.05*N, running_var ~ 1 + 0.1*U."""
    sd = module.state_dict()
    done = set()
    for name in sorted(sd.keys()):
        t = sd[name]
        if not torch.is_floating_point(t) or t.data_ptr() in done:
            continue
        done.add(t.data_ptr())  # aliased tensors (LapSRN's shared branch) are filled once, by first name
        rs = np.random.RandomState((seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        shape = tuple(t.shape)
        leaf = name.rsplit(".", 1)[-1]
        if leaf == "weight" and t.dim() >= 2:
            fan_in = int(np.prod(shape[1:]))
            if t.dim() == 4 and ".deconv." in "." + name or name.startswith("last_part"):
                fan_in = shape[0] * shape[2] * shape[3] // 4 + 1
            v = rs.standard_normal(shape) * np.sqrt(2.0 / fan_in) * gain
        elif leaf == "weight" and ".bn." in "." + name:
            v = 1.0 + 0.1 * rs.standard_normal(shape)
        elif leaf == "weight":  # PReLU slope
            v = 0.25 + 0.05 * rs.standard_normal(shape)
        elif leaf == "running_var":
            v = 1.0 + 0.1 * rs.uniform(size=shape)
        elif leaf == "running_mean":
            v = 0.1 * rs.standard_normal(shape)
        else:  # biases
            v = 0.05 * rs.standard_normal(shape)
        t.copy_(torch.from_numpy(np.asarray(v, dtype=np.float32)))
    return module


def rand(shape, seed, lo=0.0, hi=1.0):
    """Uniform [lo,hi) float32 tensor from the same platform-independent generator."""
    rs = np.random.RandomState(seed)
    return torch.from_numpy(rs.uniform(lo, hi, size=shape).astype(np.float32))


def randn(shape, seed, std=1.0):
    rs = np.random.RandomState(seed)
    return torch.from_numpy((rs.standard_normal(shape) * std).astype(np.float32))


def tensor_checksums(named_tensors):
    """{name: (sum, l2)} as a float64 array [n,2] + the name list — compact gradient fingerprints."""
    names = sorted(named_tensors.keys())
    arr = np.zeros((len(names), 2), dtype=np.float64)
    for i, n in enumerate(names):
        t = named_tensors[n].detach().double()
        arr[i, 0] = float(t.sum())
        arr[i, 1] = float(t.pow(2).sum().sqrt())
    return names, arr
