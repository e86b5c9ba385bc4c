"""Input side: the RULSTM feature reader's frame logic (datasets/reader_fns.py:41-157) over a dict store,
and the batch assembly into the loader layout.  CPU only."""
import numpy as np
import pytest
import torch

from afft_amd.datasets.reader_fns import DictStore, EpicRULSTMFeatsReader, FeatureBatcher


def _store(video, frames, C, scale=1.0):
    return DictStore({f"{video}_frame_{f:010d}.jpg".encode(): (np.arange(C, dtype=np.float32) * scale + f).tobytes()
                      for f in frames})


def test_frame_ids_match_reference_rule():
    fr = EpicRULSTMFeatsReader.frame_ids(1.0, 2.0, 30.0)      # (30, 60] at 30 fps
    assert fr[0] == 31 and fr[-1] == 60 and len(fr) == 30
    fr = EpicRULSTMFeatsReader.frame_ids(-0.2, 0.1, 30.0)     # ids below 1 are clamped to the lowest valid id
    assert fr.min() == 1 and fr[-1] == 3
    with pytest.raises(AssertionError):
        EpicRULSTMFeatsReader.frame_ids(-1.0, 0.0, 30.0)


def test_missing_frames_search_back_only_then_zeros():
    C = 8
    rd = EpicRULSTMFeatsReader(_store("P01_101", [10, 11, 14, 40], C), warn_if_using_closeby_frame=False)
    fmt = "P01_101_frame_{:010d}.jpg"
    x = rd.read_representations([10, 12, 13, 14, 23, 24, 39], rd.stores[0], fmt)
    assert x.shape == (7, 1, 1, C) and x.dtype == torch.float32
    base = torch.arange(C, dtype=torch.float32)
    got = x[:, 0, 0, :]
    assert torch.equal(got[0], base + 10)
    assert torch.equal(got[1], base + 11) and torch.equal(got[2], base + 11)     # closest EARLIER frame, never 14
    assert torch.equal(got[3], base + 14)
    assert torch.equal(got[4], base + 14)                                        # 23 - 9 = 14: still inside the radius
    assert torch.equal(got[5], torch.zeros(C))                                   # 24 - 9 = 15 > 14: zeros
    assert torch.equal(got[6], torch.zeros(C))                                   # 40 is in the future of 39
    with pytest.raises(AssertionError):
        rd.read_representations([100, 101], rd.stores[0], fmt)


def test_reader_concats_stores_and_converts_audio_fps():
    C = 4
    rgb = _store("P01_101", range(1, 200), C)
    audio = _store("P01_101", range(1, 400), C, scale=2.0)
    rd = EpicRULSTMFeatsReader([rgb, audio], ["/data/rgb_lmdb", "/data/audio_lmdb"], warn_if_using_closeby_frame=False)
    feat, _, _, _ = rd("/videos/P01_101.MP4", 1.0, 2.0, 30.0)
    assert feat.shape == (30, 1, 1, 2 * C)
    base = torch.arange(C, dtype=torch.float32)
    assert torch.equal(feat[0, 0, 0, :C], base + 31)
    assert torch.equal(feat[0, 0, 0, C:], base * 2 + round(31 / 30.0 * 50.0))      # epic-100 name -> 50 fps original video
    assert torch.equal(feat[-1, 0, 0, C:], base * 2 + 100)
    with pytest.raises(ValueError):
        EpicRULSTMFeatsReader._get_orig_video_fps("P01_1")


def test_feature_batcher_layout_and_ragged_last_batch():
    dims = {"rgb": 6, "flow": 4}
    fb = FeatureBatcher(dims, batch=3, T=5, device="cpu")
    clips = [{m: torch.full((5, 1, 1, C), float(10 * b + i)) for i, (m, C) in enumerate(dims.items())} for b in range(3)]
    out = fb.collate(clips)
    assert out["rgb"].shape == (3, 5, 6, 1, 1, 1) and out["flow"].shape == (3, 5, 4, 1, 1, 1)
    assert float(out["rgb"][2].mean()) == 20.0 and float(out["flow"][1].mean()) == 11.0
    out2 = fb.collate(clips[:2])                                  # ragged last batch
    assert out2["rgb"].shape[0] == 2
    with pytest.raises(AssertionError):
        fb.collate([{m: torch.zeros(4, C) for m, C in dims.items()}])


def test_reader_matches_reference_golden_bit_exact():
    """tests/golden/r0_reader.npz = the reference's EpicRULSTMFeatsReader (datasets/reader_fns.py:41-157) run over dict-backed
    fake LMDB environments on closed-form stores with holes (a 9-frame back-search, a 12-frame gap -> zeros, ids < 1
    clamped) and an 'audio' store in the original video's 50 fps: our reader returns the same bytes."""
    import logging
    import os
    import numpy as np
    from closed_form import reader_stores
    from afft_amd.datasets.reader_fns import DictStore, EpicRULSTMFeatsReader
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "r0_reader.npz"))
    vid, stores, queries = reader_stores()
    logging.disable(logging.CRITICAL)
    try:
        for tag, names in (("rgb", ["/data/rgb_lmdb"]), ("rgb_audio", ["/data/rgb_lmdb", "/data/audio_lmdb"])):
            rd = EpicRULSTMFeatsReader([DictStore(stores["audio" if "audio" in n else "rgb"]) for n in names], store_names=names)
            for qi, (a, b) in enumerate(queries):
                feat, _, _, _ = rd(f"/videos/{vid}.MP4", a, b, 30.0, None)
                want = z[f"{tag}:{qi}"]
                assert feat.shape == want.shape and feat.dtype == torch.float32
                assert np.array_equal(feat.numpy(), want), (tag, qi)
    finally:
        logging.disable(logging.NOTSET)


def test_open_lmdb_binding_over_a_stand_in_lmdb_module(monkeypatch):
    """open_lmdb: the LMDB binding itself (readonly, lock-free environment; one read transaction per get), exercised with a
    stand-in `lmdb` module of the same interface (the package is not in the image)."""
    import sys
    import types
    from afft_amd.datasets import reader_fns as R
    data = {b"k1": b"abcd"}
    opened = {}

    class _Txn:
        def __enter__(self):
            return self

        def __exit__(self, *a):
            return False

        def get(self, k):
            return data.get(k)

    class _Env:
        def begin(self):
            return _Txn()

    def _open(path, readonly=False, lock=True):
        opened.update(path=path, readonly=readonly, lock=lock)
        return _Env()
    monkeypatch.setitem(sys.modules, "lmdb", types.SimpleNamespace(open=_open))
    st = R.open_lmdb("/data/x_lmdb")
    assert opened == dict(path="/data/x_lmdb", readonly=True, lock=False)
    assert st.get(b"k1") == b"abcd" and st.get(b"nope") is None
