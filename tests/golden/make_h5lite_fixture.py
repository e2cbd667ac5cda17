"""Writes tests/golden/h5lite_fixture.h5 (+ .npz with the same arrays): a small HDF5 file of the build's OWN data that
exercises what framework/h5lite.py reads - many root datasets (several symbol-table nodes / B-tree entries), nested groups,
1-3-D float64 / float32 / int64 / int32 / uint8, big-endian data, an enumeration (booleans), fixed-length strings, a scalar and a compact dataset, a never-written dataset,
chunked datasets (plain, gzip, shuffle + gzip + fletcher32 with ragged edge chunks, one with chunks never written) - plus
one lzf-compressed dataset that h5lite must refuse BY NAME, and two more files it must refuse on opening:
h5lite_unsupported_latest.h5 (libver="latest": superblock 3) and h5lite_unsupported_newgroup.h5 (a group with link
messages).  Run with an interpreter that has h5py
(`/opt/conda/bin/python3.9 tests/golden/make_h5lite_fixture.py`); the tests read the committed files."""
import os

import h5py
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(7)
data = {}
for i in range(40):                                   # > 2 * group leaf K entries: the root B-tree gets several nodes
    data[f"F{i:02d}"] = rng.normal(size=(5, 7)).astype(np.float64 if i % 2 else np.float32)
data["KLEV"] = np.array([137], dtype=np.int64)
data["KLON"] = np.array([100], dtype=np.int32)
data["CUBE"] = rng.integers(0, 255, size=(3, 4, 5)).astype(np.uint8)
data["BIG_ENDIAN"] = np.arange(12, dtype=">f8").reshape(3, 4)
data["LOGICAL"] = np.array([True, False, True])          # h5py: enumeration over int8
data["NAME"] = np.array([b"cloudsc2", b"hip"], dtype="S8")  # fixed-length strings
data["grp/inner/T"] = rng.normal(size=(6,))
data["grp/Q"] = rng.normal(size=(2, 3)).astype(np.float32)
path = os.path.join(HERE, "h5lite_fixture.h5")
with h5py.File(path, "w", libver="earliest") as f:
    for k, v in data.items():
        f.create_dataset(k, data=v)
    f.create_dataset("SCALAR", data=np.float64(3.5))
    f.create_dataset("NEVER_WRITTEN", shape=(4, 2), dtype=np.float64)
    dcpl = h5py.h5p.create(h5py.h5p.DATASET_CREATE)
    dcpl.set_layout(h5py.h5d.COMPACT)
    space = h5py.h5s.create_simple((3,))
    dset = h5py.h5d.create(f.id, b"COMPACT", h5py.h5t.NATIVE_INT32, space, dcpl)
    dset.write(h5py.h5s.ALL, h5py.h5s.ALL, np.array([7, 8, 9], dtype=np.int32))
    data["CHUNKED_GZIP"] = rng.normal(size=(64, 64))
    f.create_dataset("CHUNKED_GZIP", data=data["CHUNKED_GZIP"], chunks=(16, 16), compression="gzip")
    data["CHUNKED_PLAIN"] = rng.normal(size=(10, 7)).astype(np.float32)
    f.create_dataset("CHUNKED_PLAIN", data=data["CHUNKED_PLAIN"], chunks=(4, 3))            # ragged edge chunks
    data["CHUNKED_SHUFFLE"] = rng.integers(-1000, 1000, size=(5, 33, 9)).astype(np.int64)
    f.create_dataset("CHUNKED_SHUFFLE", data=data["CHUNKED_SHUFFLE"], chunks=(2, 8, 9), compression="gzip", shuffle=True,
                     fletcher32=True)
    sparse = f.create_dataset("CHUNKED_SPARSE", shape=(12, 12), dtype=np.float64, chunks=(4, 4))
    sparse[4:8, 8:12] = 2.5                                                                  # one chunk of nine written
    data["CHUNKED_SPARSE"] = np.zeros((12, 12))
    data["CHUNKED_SPARSE"][4:8, 8:12] = 2.5
    f.create_dataset("CHUNKED_LZF", data=rng.normal(size=(8, 8)), chunks=(4, 4), compression="lzf")   # refused: filter 32000
with h5py.File(os.path.join(HERE, "h5lite_unsupported_latest.h5"), "w", libver="latest") as f:
    f.create_dataset("X", data=np.arange(4.0))
with h5py.File(os.path.join(HERE, "h5lite_unsupported_newgroup.h5"), "w", libver="earliest") as f:
    f.create_group("tracked", track_order=True).create_dataset("X", data=np.arange(4.0))
data["SCALAR"] = np.float64(3.5)
data["NEVER_WRITTEN"] = np.zeros((4, 2))
data["COMPACT"] = np.array([7, 8, 9], dtype=np.int32)
np.savez(os.path.join(HERE, "h5lite_fixture.npz"), **{k.replace("/", "__"): v for k, v in data.items()})
print(path, os.path.getsize(path), "bytes,", len(data) + 1, "datasets")
