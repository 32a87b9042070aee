"""
CPU tests of the minimal HDF5 writer/reader (esc_gnn_amd/hdf5_min.py) for the reference's operator-file layout
(generate_ellipsoid_dataset.py:1185-1206; reader escgnn_data_classes.py:135-143).  No HDF5 library exists in this image:
the checks are the published lookup3 test vectors, spec-level structure, and a round trip through the module's reader.
"""
import os
import struct
import numpy as np
import pytest

from esc_gnn_amd import hdf5_min as h5


def test_lookup3_vectors():
    # from Bob Jenkins' lookup3.c self-test (driver5); both forms: plain Python and libescat's C
    assert h5.lookup3_py(b"") == 0xdeadbeef
    s0 = b"Four score and seven years ago"
    assert h5.lookup3_py(s0, 0) == 0x17770551 and h5.lookup3_py(s0, 1) == 0xcd628161
    import numpy as np
    rng = np.random.default_rng(0)
    for n in (0, 1, 11, 12, 13, 24, 25, 1000):
        d = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        assert h5.lookup3(d, 7) == h5.lookup3_py(d, 7)
    assert h5.lookup3(b"") == 0xdeadbeef
    s = b"Four score and seven years ago"
    assert h5.lookup3(s, 0) == 0x17770551
    assert h5.lookup3(s, 1) == 0xcd628161


def test_operator_file_round_trip(tmp_path, bg):
    path = str(tmp_path / "pq.h5")
    graphs = []
    for ci in range(3):
        p = f"g{ci}_"
        d = {"P": {"indices": bg[p + "P_indices"], "values": bg[p + "P_values_f16"], "size": bg[p + "P_size"]},
             "Q": {"indices": bg[p + "Q_indices"], "values": bg[p + "Q_values_f32"], "size": bg[p + "Q_size"]},
             "edge_index": bg[p + "edge_index"]}
        graphs.append((ci, d))
    h5.write_operator_file(path, graphs)
    assert not os.path.exists(path + ".tmp")
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 2 and raw[9] == 8 and raw[10] == 8
    eof, root = struct.unpack("<QQ", raw[28:44])
    assert eof == len(raw) and raw[root:root + 4] == b"OHDR"
    f = h5.H5Reader(path)
    assert set(f.keys()) == {"P", "Q", "edge_index"}
    assert set(f["P"].keys()) == {"0", "1", "2"}
    for ci in range(3):
        p = f"g{ci}_"
        grp = f["P"][str(ci)]
        assert grp["indices"].dtype == np.int64 and grp["indices"].shape == bg[p + "P_indices"].shape
        assert np.array_equal(grp["indices"][...], bg[p + "P_indices"])
        assert grp["values"].dtype == np.float16 and np.array_equal(grp["values"][...], bg[p + "P_values_f16"])
        assert tuple(grp["size"][...].tolist()) == tuple(bg[p + "P_size"])
        q = f["Q"][str(ci)]
        assert q["values"].dtype == np.float32 and np.array_equal(q["values"][...], bg[p + "Q_values_f32"])
        assert np.array_equal(q["indices"][...], bg[p + "Q_indices"])
        assert np.array_equal(f["edge_index/%d/values" % ci][...], bg[p + "edge_index"])
    # what the reference's loader does with it (escgnn_data_classes.py:137-142)
    import torch
    g = f["Q"]["1"]
    Q = torch.sparse_coo_tensor(torch.from_numpy(g["indices"][...]), torch.from_numpy(g["values"][...]),
                                tuple(g["size"][...].tolist()), dtype=torch.float32)
    assert Q.shape == (192, 192) and Q._nnz() == bg["g1_Q_values_f32"].shape[0]
    f.close()


def test_corruption_is_detected(tmp_path):
    path = str(tmp_path / "x.h5")
    with h5.H5Writer(path) as w:
        w.write("a/b", np.arange(6, dtype=np.int64).reshape(2, 3))
        w.write("a/empty", np.zeros((0,), dtype=np.float32))
        with pytest.raises(ValueError):
            w.write("a/b", np.zeros(1))
    f = h5.H5Reader(path)
    assert np.array_equal(f["a"]["b"][...], np.arange(6).reshape(2, 3)) and f["a"]["empty"][...].shape == (0,)
    f.close()
    raw = bytearray(open(path, "rb").read())
    root = struct.unpack("<Q", raw[36:44])[0]
    raw[root + 12] ^= 0xFF
    open(path, "wb").write(raw)
    with pytest.raises(ValueError):
        h5.H5Reader(path)["a"]
