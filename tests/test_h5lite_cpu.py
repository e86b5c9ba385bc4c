"""afft_amd.h5lite -- the HDF5 subset of the reference's logits files (test.py:20-31) written and read without h5py.

The self-consistency tests run everywhere.  Where an interpreter WITH h5py exists (this image: /opt/conda/bin/python3.9, h5py
3.3 on libhdf5 1.10; override with AFFT_H5PY_PYTHON) the file format itself is pinned against the real library: h5py reads
what h5lite writes and finds the reference's dataset properties, h5py appends to it the reference's way, h5lite reads and
extends what h5py wrote, and the HDF5 command-line tools parse it."""
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

from afft_amd import h5lite

KEY = "logits/action_all-fused"
H5PY_PYTHON = os.environ.get("AFFT_H5PY_PYTHON", "/opt/conda/bin/python3.9")

# what the other interpreter runs: inspect a file / append the reference's way (store_append_h5, test.py:20-31)
_HELPER = r'''
import sys, json, numpy as np, h5py
mode, path = sys.argv[1], sys.argv[2]
if mode == "inspect":
    out = {}
    with h5py.File(path, "r") as f:
        def visit(name, obj):
            if isinstance(obj, h5py.Dataset):
                a = np.asarray(obj[...])
                out[name] = {"shape": list(obj.shape), "maxshape": list(obj.maxshape), "dtype": str(obj.dtype),
                             "chunks": list(obj.chunks) if obj.chunks else None, "compression": obj.compression,
                             "compression_opts": obj.compression_opts, "sum": float(a.astype(np.float64).sum()),
                             "sumsq": float((a.astype(np.float64) ** 2).sum())}
        f.visititems(visit)
    print(json.dumps(out))
else:
    key, n, seed, cols = sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    val = np.random.default_rng(seed).standard_normal((n, cols)).astype(np.float32)
    with h5py.File(path, "a") as fout:
        if key not in fout:
            fout.create_dataset(key, data=val, compression="gzip", compression_opts=9, chunks=True, maxshape=(None,) + val.shape[1:])
        else:
            fout[key].resize((fout[key].shape[0] + val.shape[0],) + val.shape[1:])
            fout[key][-val.shape[0]:, ...] = val
    print("ok")
'''


def _have_h5py() -> bool:
    if not os.path.exists(H5PY_PYTHON):
        return False
    r = subprocess.run([H5PY_PYTHON, "-c", "import h5py"], capture_output=True)
    return r.returncode == 0


needs_h5py = pytest.mark.skipif(not _have_h5py(), reason=f"no interpreter with h5py at {H5PY_PYTHON}")


def _h5py(tmp_path, *args):
    helper = tmp_path / "h5helper.py"
    if not helper.exists():
        helper.write_text(_HELPER)
    r = subprocess.run([H5PY_PYTHON, str(helper)] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stdout.strip()


def _batch(seed, n, cols=3806):
    return np.random.default_rng(seed).standard_normal((n, cols)).astype(np.float32)


def test_write_read_round_trip_types_groups_and_ragged_chunks(tmp_path):
    rng = np.random.default_rng(1)
    data = {KEY: _batch(0, 70), "meta/idx": rng.integers(-5, 5, (10, 3, 4)).astype(np.int32),
            "meta/deep/er/u8": rng.integers(0, 255, (1000,)).astype(np.uint8), "top64": rng.standard_normal((5, 2)),
            "empty": np.zeros((0, 7), np.float32)}
    p = str(tmp_path / "a.h5")
    h5lite.write(p, data, chunks={"meta/idx": (4, 2, 3)})      # chunk grid that divides no dimension
    back = h5lite.read(p)
    assert set(back) == set(data)
    for k, v in data.items():
        assert back[k].dtype == v.dtype and np.array_equal(back[k], v), k
    f = h5lite.File(p)
    info = f.info(KEY)
    f.close()
    assert info["shape"] == (70, 3806) and info["maxshape"] == (None, 3806) and info["compression"] == "gzip" \
        and info["compression_opts"] == 9 and info["chunks"][1] == 3806
    with pytest.raises(ValueError):
        h5lite.write(p, {"x": np.float32(1.0)})
    with pytest.raises(ValueError):
        h5lite.File(__file__)


def test_append_grows_in_place_like_the_reference_loop(tmp_path):
    """store_append_h5 called batch after batch: ragged batches, a last partial chunk that is re-stored, many chunks (a two-level
    chunk index: > 64 chunks), a second dataset appearing later."""
    p = str(tmp_path / "run.h5")
    parts = []
    for i, n in enumerate([64, 64, 13, 1, 64, 7]):
        b = _batch(10 + i, n, 96)
        parts.append(b)
        h5lite.append(p, {KEY: b})
        assert np.array_equal(h5lite.read(p, KEY), np.concatenate(parts)), i
    small = str(tmp_path / "many.h5")
    h5lite.write(small, {KEY: _batch(3, 2, 5)}, chunks={KEY: (2, 5)})
    rows = [_batch(3, 2, 5)]
    for i in range(5):
        b = _batch(40 + i, 61, 5)           # 30.5 chunks per append -> 154 chunks, index two levels deep
        rows.append(b)
        h5lite.append(small, {KEY: b})
    assert np.array_equal(h5lite.read(small, KEY), np.concatenate(rows))
    f = h5lite.File(small)
    assert np.array_equal(f.rows(KEY, 123), np.concatenate(rows)[123:])
    f.close()
    h5lite.append(p, {"logits/action_rgb": parts[0]})          # a new key in an existing file
    back = h5lite.read(p)
    assert np.array_equal(back[KEY], np.concatenate(parts)) and np.array_equal(back["logits/action_rgb"], parts[0])
    with pytest.raises(ValueError):
        h5lite.append(p, {KEY: _batch(0, 3, 95)})


def test_evaluate_store_append_writes_hdf5(tmp_path):
    from afft_amd import evaluate as E
    a, b = _batch(1, 9, 11), _batch(2, 4, 11)
    path = E.store_append({KEY: a}, str(tmp_path / "out"), "logits.h5")
    assert E.store_append({KEY: b}, str(tmp_path / "out"), "logits.h5") == path
    with open(path, "rb") as fh:
        assert fh.read(8) == b"\x89HDF\r\n\x1a\n"
    assert np.array_equal(E.load_logits(path, KEY), np.concatenate([a, b]))
    assert list(E.load_logits(path)) == [KEY]


def test_challenge_loader_reads_the_stored_logits(tmp_path):
    """challenge.gen_load_resfiles (challenge.py:79-91): the 'test*h5' files of a run directory as {leaf key: array}."""
    from afft_amd import challenge as CH, evaluate as E
    a = _batch(5, 6, 13)
    E.store_append({KEY: a}, str(tmp_path), "test_run.h5")
    E.store_append({KEY: a[:2]}, str(tmp_path), "test_run.h5")
    (tmp_path / "notes.txt").write_text("not a result file")
    res = list(CH.gen_load_resfiles(str(tmp_path)))
    assert len(res) == 1 and list(res[0]) == [KEY] and np.array_equal(res[0][KEY], np.concatenate([a, a[:2]]))
    with pytest.raises(ValueError):
        next(CH.gen_load_resfiles(str(tmp_path / "nothing_here")))


@needs_h5py
def test_h5py_reads_what_h5lite_writes_and_appends_to_it(tmp_path):
    p = str(tmp_path / "lite.h5")
    a, b = _batch(0, 70), _batch(1, 64)
    h5lite.write(p, {KEY: a, "meta/idx": np.arange(24, dtype=np.int32).reshape(2, 3, 4)})
    info = json.loads(_h5py(tmp_path, "inspect", p))
    d = info[KEY]
    assert d["shape"] == [70, 3806] and d["maxshape"] == [None, 3806] and d["dtype"] == "float32"
    assert d["compression"] == "gzip" and d["compression_opts"] == 9 and d["chunks"][1] == 3806
    assert d["sum"] == float(a.astype(np.float64).sum()) and d["sumsq"] == float((a.astype(np.float64) ** 2).sum())
    assert info["meta/idx"]["dtype"] == "int32" and info["meta/idx"]["sum"] == 276.0
    h5lite.append(p, {KEY: b})                                   # grown in place by h5lite ...
    d = json.loads(_h5py(tmp_path, "inspect", p))[KEY]
    ab = np.concatenate([a, b])
    assert d["shape"] == [134, 3806] and d["sum"] == float(ab.astype(np.float64).sum())
    _h5py(tmp_path, "append", p, KEY, 50, 7, 3806)               # ... then by libhdf5 (resize + assignment, the reference's code)
    abc = np.concatenate([ab, _batch(7, 50)])
    assert np.array_equal(h5lite.read(p, KEY), abc)
    h5lite.append(p, {KEY: a[:5]})                               # ... and by h5lite again, on the file libhdf5 modified
    d = json.loads(_h5py(tmp_path, "inspect", p))[KEY]
    full = np.concatenate([abc, a[:5]])
    assert d["shape"] == [189, 3806] and d["sum"] == float(full.astype(np.float64).sum())
    assert np.array_equal(h5lite.read(p, KEY), full)


@needs_h5py
def test_h5lite_reads_and_extends_what_h5py_writes(tmp_path):
    """A file made entirely by the reference's store_append_h5 under h5py (its own chunk guess: 2-D chunk grid)."""
    p = str(tmp_path / "ref.h5")
    parts = []
    for i, n in enumerate((64, 64, 23)):
        _h5py(tmp_path, "append", p, KEY, n, 100 + i, 3806)
        parts.append(_batch(100 + i, n))
    exp = np.concatenate(parts)
    f = h5lite.File(p)
    info = f.info(KEY)
    got = f[KEY]
    f.close()
    assert info["shape"] == (151, 3806) and info["maxshape"] == (None, 3806) and info["compression_opts"] == 9
    assert np.array_equal(got, exp)
    h5lite.append(p, {KEY: parts[0][:9]})
    d = json.loads(_h5py(tmp_path, "inspect", p))[KEY]
    assert d["shape"] == [160, 3806] and d["sum"] == float(np.concatenate([exp, parts[0][:9]]).astype(np.float64).sum())


@needs_h5py
def test_hdf5_command_line_tools_parse_the_file(tmp_path):
    tool = os.path.join(os.path.dirname(H5PY_PYTHON), "h5dump")
    if not shutil.which(tool):
        pytest.skip("no h5dump")
    p = str(tmp_path / "lite.h5")
    h5lite.write(p, {KEY: _batch(0, 3, 4)})
    r = subprocess.run([tool, "-H", "-p", p], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert 'DATASET "action_all-fused"' in out and "H5T_IEEE_F32LE" in out and "H5S_UNLIMITED" in out
    assert "COMPRESSION DEFLATE { LEVEL 9 }" in out and "CHUNKED" in out
