"""Input side of the path: pre-extracted RULSTM features -> the loader layout ``(B, T, C, 1, 1, 1)``
fp32 on the GPU.

``EpicRULSTMFeatsReader`` mirrors the reference's reader (datasets/reader_fns.py:41-157): one key per frame,
``"<video>_frame_{:010d}.jpg"``, value = the raw float32 bytes of the feature vector; a frame that is not stored is
replaced by the closest EARLIER stored frame within 9 frames (never a later one: anticipation), else by zeros; frame ids
below 1 are clamped to the lowest valid id; audio / poses stores are indexed in the original video's frame rate.
The store is anything with ``get(key: bytes) -> Optional[bytes]``: an LMDB transaction when the ``lmdb`` package is
installed (``open_lmdb``; it is not in this image), a dict in the tests.

``FeatureBatcher`` is what replaces the default collate + ``.to(device)`` of the reference's loop (common/runner.py:
245-256): clips are written straight into ONE pinned staging buffer per modality and leave with one non-blocking
host-to-device copy each; ``ZeroMaskRULSTMFeats`` and MixUp then run on the device (afft_amd.common.transforms / mixup).
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

FRAME_FORMAT = "{}_frame_{{:010d}}.jpg"


class DictStore:
    """key -> bytes store with the one method the reader needs (tests, in-memory caches)."""

    def __init__(self, items: Dict[bytes, bytes]):
        self.items = items

    def get(self, key: bytes) -> Optional[bytes]:
        return self.items.get(key)


def open_lmdb(path):
    """LMDB environment as a store (readonly, no lock -- datasets/reader_fns.py:53).  Needs the `lmdb` package."""
    import lmdb  # noqa: PLC0415  (absent from the build image; present wherever the reference's datasets are)

    env = lmdb.open(str(path), readonly=True, lock=False)

    class _Txn:
        def get(self, key: bytes) -> Optional[bytes]:
            with env.begin() as e:
                return e.get(key)

    return _Txn()


class EpicRULSTMFeatsReader:
    def __init__(self, stores: Union[object, Sequence[object]], store_names: Optional[Sequence[str]] = None,
                 warn_if_using_closeby_frame: bool = True):
        """stores: one store or a list (their features are concatenated, reader_fns.py:45-47); store_names: the LMDB
        paths they came from -- a name containing 'audio' or 'poses' switches that store to the original video's frame
        rate (reader_fns.py:127-129)."""
        self.stores = list(stores) if isinstance(stores, (list, tuple)) else [stores]
        self.names = list(store_names) if store_names is not None else [""] * len(self.stores)
        assert len(self.names) == len(self.stores)
        self.warn_if_using_closeby_frame = warn_if_using_closeby_frame

    @staticmethod
    def get_frame_rate(video_path) -> float:
        del video_path
        return 30.0

    def read_representations(self, frames, store, frame_format: str) -> torch.Tensor:
        """(len(frames), 1, 1, C) fp32; reader_fns.py:66-107."""
        features: List[Optional[np.ndarray]] = []
        for frame_id in frames:
            dd = None
            search_radius = 0
            for search_radius in range(10):      # only frames at or before the requested one
                dd = store.get(frame_format.format(int(frame_id) - search_radius).strip().encode("utf-8"))
                if dd is not None:
                    break
            if dd is not None and search_radius > 0 and self.warn_if_using_closeby_frame:
                logging.warning("Missing %s, but used %d instead", frame_format.format(int(frame_id)),
                                int(frame_id) - search_radius)
            if dd is None:
                logging.error("Missing %s, Only specific frames are stored in lmdb :(", frame_format.format(int(frame_id)))
                features.append(None)
            else:
                features.append(np.frombuffer(dd, "float32"))
        found = [el for el in features if el is not None]
        assert len(found) > 0, f"No features found in {frame_format} - {frames}"
        feats = np.array([np.zeros_like(found[0]) if el is None else el for el in features])
        return torch.as_tensor(feats[:, np.newaxis, np.newaxis, :])

    @staticmethod
    def _get_orig_video_fps(video_name: str) -> float:
        length = len(video_name.split("_")[-1])
        if length == 3:      # epic 100
            return 50.0
        if length == 2:      # epic 55
            return 59.94005994005994
        raise ValueError(f"Unkown video name format: {video_name}")

    def _convert_to_orig_video_fps(self, video_name, fps, frames):
        return np.rint(frames / fps * self._get_orig_video_fps(video_name)).astype(int)

    @staticmethod
    def frame_ids(start_sec: float, end_sec: float, fps: float) -> np.ndarray:
        """every frame in (start, end], ids below 1 clamped to the lowest valid one (reader_fns.py:115-122)"""
        start_frame, end_frame = np.floor(start_sec * fps), np.floor(end_sec * fps)
        frames = np.arange(end_frame, start_frame, -1).astype(int)[::-1]
        assert frames.max() >= 1, f"The dataset shouldnt have cases otherwise. {start_sec} {end_sec} {frames}"
        frames[frames < 1] = frames[frames >= 1].min()
        return frames

    def __call__(self, video_path, start_sec: float, end_sec: float, fps: float, df_row=None, pts_unit="sec"):
        del df_row, pts_unit
        frames = self.frame_ids(start_sec, end_sec, fps)
        video_name = Path(video_path).stem
        all_feats = []
        for store, name in zip(self.stores, self.names):
            fr = self._convert_to_orig_video_fps(video_name, fps, frames) if ("audio" in name or "poses" in name) else frames
            all_feats.append(self.read_representations(fr, store, FRAME_FORMAT.format(video_name)))
        return torch.cat(all_feats, dim=-1), {}, {}, {}


class FeatureBatcher:
    """Collates per-clip features into the loader layout on the device: for every modality ONE pinned staging buffer
    (B, T, C, 1, 1, 1) that the clips are written into directly, one non-blocking copy to the GPU per modality per batch,
    double-buffered so that batch i+1 is staged while batch i is in flight."""

    def __init__(self, modal_dims: Dict[str, int], batch: int, T: int, device, depth: int = 2):
        self.dims, self.B, self.T, self.device = dict(modal_dims), batch, T, torch.device(device)
        pin = self.device.type == "cuda"
        self.host = [{m: torch.zeros(batch, T, C, 1, 1, 1, dtype=torch.float32, pin_memory=pin) for m, C in self.dims.items()}
                     for _ in range(depth)]
        self.events = [None] * depth
        self.turn = 0

    def collate(self, clips: Sequence[Dict[str, torch.Tensor]]) -> Dict[str, torch.Tensor]:
        """clips: per sample {modality: (T, 1, 1, C) as the reader returns, or (T, C)} -> {modality: (b, T, C, 1, 1, 1)} on
        the device (b = len(clips) <= B)."""
        assert 0 < len(clips) <= self.B
        slot = self.turn
        self.turn = (self.turn + 1) % len(self.host)
        if self.events[slot] is not None:
            self.events[slot].synchronize()        # the copy that last used this staging buffer has left the host
        out = {}
        for m, C in self.dims.items():
            buf = self.host[slot][m]
            for b, clip in enumerate(clips):
                f = clip[m].reshape(-1, C)
                assert f.shape[0] == self.T, f"{m}: clip has {f.shape[0]} frames, expected {self.T}"
                buf[b, :, :, 0, 0, 0].copy_(f)
            out[m] = buf[:len(clips)].to(self.device, non_blocking=True)
        if self.device.type == "cuda":
            self.events[slot] = torch.cuda.Event()
            self.events[slot].record()
        return out
