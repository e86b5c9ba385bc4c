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
