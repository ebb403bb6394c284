"""Per-scene ReID feature blob: the data format on the input side of the hot path.

The reference stores one pickle of a CPU `torch.Tensor[2048]` per tracklet at
`reid_features/<scene>/c<cam:03d>/<id:04d>/<file>_<model>.pkl` (written at libs/reid_feature_extraction.py:176-184,
read back one by one at libs/dataset.py:298-307) and uploads them one `.cuda()` call per node (inference.py:389-400).
Here a scene is ONE file: a 4 KiB header, the (camera, id) index, and the `[N, F]` float32 matrix, page-aligned, so it
can be memory-mapped and moved to HBM in a single host-to-device copy -- straight into `build_graph`.

    layout (little endian):
      0     8   magic  b"MTMCFEAT"
      8     4   u32 version (1)
      12    4   u32 F (feature width)
      16    8   u64 N (tracklets)
      24    8   u64 offset of the index   (i32 cam[N], then i32 id[N])
      32    8   u64 offset of the matrix  (f32 [N][F], 4096-byte aligned)
      40    8   u64 total file size
      48   ..   zero padding up to 4096
    rows are sorted by (camera, id): the order in which the reference's dataset enumerates tracklets
    (libs/dataset.py:272-281: cameras in order, `np.unique(ids)` within a camera).
"""
from __future__ import annotations

import os
import pickle
import re
import struct
from typing import Iterable, Optional, Sequence, Tuple

import numpy as np
import torch

MAGIC = b"MTMCFEAT"
VERSION = 1
PAGE = 4096
_HEADER = struct.Struct("<8sIIQQQQ")


def _align(v: int, a: int = PAGE) -> int:
    return (v + a - 1) // a * a


def write(path: str, cams: Sequence[int], ids: Sequence[int], feats) -> None:
    """Write a scene blob.  feats: [N, F] float32 (numpy or CPU/GPU torch); rows are re-ordered by (camera, id)."""
    cams = np.asarray(cams, dtype=np.int64)
    ids = np.asarray(ids, dtype=np.int64)
    if isinstance(feats, torch.Tensor):
        feats = feats.detach().cpu().numpy()
    feats = np.ascontiguousarray(feats, dtype=np.float32)
    if feats.ndim != 2 or feats.shape[0] != cams.size or ids.size != cams.size:
        raise ValueError("feature_store.write: feats must be [N, F] with one (camera, id) per row")
    order = np.lexsort((ids, cams))
    if order.size > 1:
        key = cams[order] * (1 << 32) + ids[order]
        if np.any(key[1:] == key[:-1]):
            raise ValueError("feature_store.write: duplicate (camera, id)")
    n, f = feats.shape
    idx_off = PAGE
    mat_off = _align(idx_off + 8 * n)
    total = mat_off + 4 * n * f
    tmp = path + ".tmp"
    with open(tmp, "wb") as fh:
        fh.write(_HEADER.pack(MAGIC, VERSION, f, n, idx_off, mat_off, total).ljust(PAGE, b"\0"))
        fh.write(cams[order].astype("<i4").tobytes())
        fh.write(ids[order].astype("<i4").tobytes())
        fh.write(b"\0" * (mat_off - idx_off - 8 * n))
        fh.write(feats[order].astype("<f4", copy=False).tobytes())
    os.replace(tmp, path)


def convert(reid_root: str, scene: str, file: str, model_name: str, out_path: str) -> Tuple[int, int]:
    """Pack the reference's per-tracklet pickles of one scene (`<reid_root>/<scene>/c0NN/<id>/<file>_<model>.pkl`)
    into one blob.  Returns (N, F)."""
    base = os.path.join(reid_root, scene)
    cams, ids, rows = [], [], []
    name = file + "_" + model_name + ".pkl"
    for cdir in sorted(os.listdir(base)):
        m = re.fullmatch(r"c(\d+)", cdir)
        if not m or not os.path.isdir(os.path.join(base, cdir)):
            continue
        for idir in sorted(os.listdir(os.path.join(base, cdir))):
            fp = os.path.join(base, cdir, idir, name)
            if not (idir.isdigit() and os.path.isfile(fp)):
                continue
            with open(fp, "rb") as fh:
                t = pickle.load(fh)                      # a CPU torch.Tensor[F], as the reference dumped it
            rows.append(torch.as_tensor(t).detach().to(torch.float32).reshape(-1).numpy())
            cams.append(int(m.group(1)))
            ids.append(int(idir))
    if not rows:
        raise FileNotFoundError(f"feature_store.convert: no '{name}' under {base}")
    feats = np.stack(rows)
    write(out_path, cams, ids, feats)
    return feats.shape


class FeatureStore:
    """Memory-mapped view of a scene blob."""

    def __init__(self, path: str):
        size = os.path.getsize(path)
        with open(path, "rb") as fh:
            head = fh.read(_HEADER.size)
        if len(head) < _HEADER.size:
            raise ValueError(f"{path}: not a feature blob (too short)")
        magic, version, f, n, idx_off, mat_off, total = _HEADER.unpack(head)
        if magic != MAGIC:
            raise ValueError(f"{path}: bad magic {magic!r}")
        if version != VERSION:
            raise ValueError(f"{path}: unsupported version {version}")
        if total != size or mat_off % PAGE or idx_off + 8 * n > mat_off or mat_off + 4 * n * f != total:
            raise ValueError(f"{path}: header does not match the file ({size} bytes): truncated or corrupt")
        self.path, self.n, self.f = path, int(n), int(f)
        self.cams = np.memmap(path, dtype="<i4", mode="r", offset=idx_off, shape=(self.n,))
        self.ids = np.memmap(path, dtype="<i4", mode="r", offset=idx_off + 4 * self.n, shape=(self.n,))
        self.feats = np.memmap(path, dtype="<f4", mode="r", offset=mat_off, shape=(self.n, self.f))
        self._key = self.cams.astype(np.int64) * (1 << 32) + self.ids.astype(np.int64)      # ascending by construction

    def __len__(self) -> int:
        return self.n

    def rows(self, cams: Iterable[int], ids: Iterable[int]) -> np.ndarray:
        """Row of every (camera, id); KeyError on a tracklet the blob does not hold (the reference would fail on the
        missing pickle, libs/dataset.py:301-307)."""
        want = np.asarray(list(cams), dtype=np.int64) * (1 << 32) + np.asarray(list(ids), dtype=np.int64)
        pos = np.searchsorted(self._key, want)
        pos = np.minimum(pos, self.n - 1)
        bad = self._key[pos] != want
        if np.any(bad):
            k = int(want[np.argmax(bad)])
            raise KeyError(f"{self.path}: no features for camera {k >> 32}, id {k & 0xffffffff}")
        return pos

    def to_device(self, device, cams: Optional[Iterable[int]] = None, ids: Optional[Iterable[int]] = None,
                  pin: bool = False) -> torch.Tensor:
        """[N, F] float32 on `device` in ONE host-to-device copy (whole scene), or the given tracklets in the given
        order (gathered on the host first, still one copy)."""
        import warnings
        if cams is None:
            with warnings.catch_warnings():                 # read-only mapping: the tensor is only ever copied from
                warnings.simplefilter("ignore", UserWarning)
                host = torch.from_numpy(np.ascontiguousarray(self.feats))
        else:
            host = torch.from_numpy(self.feats[self.rows(cams, ids)])
        if pin and torch.cuda.is_available():
            host = host.pin_memory()
        return host.to(device, non_blocking=pin)

    def tracklets(self):
        """(cam_ids, ids) in dataset order -- what inference.py:389-400 collects as cam_ids_nodes / node_labels_g."""
        return np.asarray(self.cams).copy(), np.asarray(self.ids).copy()
