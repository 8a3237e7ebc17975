"""Training batches straight from sequence folders, assembled on the GPU -- the counterpart of the reference's
``dataset.SfMDataset`` (train / validation phases, dataset.py:133-333 set-up, 335-460 ``__getitem__``) plus the ``DataLoader``
that batches its samples (train.py:168-178, 254-270), without OpenCV / plyfile / albumentations and without a host image.

    per folder (once)   reader.load_sequence  (or the reference's own precompute file, ``use_store_data``: dataset.py:320-331)
                        -> scatter.SequenceScatter: point cloud, mask, visibility, per-view matrices resident in HBM
    per sample          utils.generating_pos_and_increment (host RNG, same calls as the reference)
    per batch           SequenceScatter.training_batch -> the 14 non-image tensors, reader.FrameDecoder -> the two colour tensors

What the reference does per sample and this does not: the albumentations colour / blur / noise augmentations (train.py:121-144;
out of scope) -- the colour tensors are the normalised frames, i.e. the reference's validation-phase tensors.
A sample whose sparse depth masks come out empty is redrawn as in dataset.py:372-376.
"""

import os
import pickle
import queue
import random
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import reader, scatter, utils


def read_precompute_file(path):
    """The list the reference pickles (dataset.py:310-319) as a dictionary of its fourteen entries."""
    names = ("crop_positions", "selected_indexes", "visible_view_indexes", "point_cloud", "intrinsic_matrix", "mask_boundary",
             "view_indexes_per_point", "extrinsics", "projection", "clean_point_list", "downsampling", "network_downsampling",
             "inlier_percentage", "estimated_scale")
    with open(str(path), "rb") as f:
        entries = pickle.load(f)
    if len(entries) != len(names):
        raise ValueError("%s does not hold the %d entries of a precompute file" % (path, len(names)))
    return dict(zip(names, entries))


def _by_folder_name(table, folder):
    """The reference keys its dictionaries by str(folder) on the machine that wrote the file; match by the folder's name."""
    if folder in table:
        return table[folder]
    name = os.path.basename(os.path.normpath(folder))
    for key, value in table.items():
        if os.path.basename(os.path.normpath(str(key))) == name:
            return value
    raise KeyError("no precomputed entry for sequence %s" % folder)


class TrainingBatches(object):
    """Iterable over training batches (dictionaries keyed as ``synthetic.BATCH_KEYS``, everything on the device).

    folder_list           sequence folders (``<root>/bag_x/_start_...``), as utils.get_parent_folder_names returns them
    adjacent_range        (min, max) frame gap of a pair (train.py --adjacent_range)
    precompute_path       the reference's precompute pickle (optional): crop window, matrices, scale and contaminated-point lists
                          of every sequence as the reference stored them.  Without it the folders are read with
                          reader.load_sequence; inlier_percentage (0.99 in train.py) then runs the contaminated-point filter
                          on each folder's frames, None skips it.
    image_file_names      the sample list (default: every ``0*.jpg`` of the folders, sorted, as utils.get_color_file_names)
    num_iter              samples per epoch (dataset.py:137, 333); default: len(image_file_names)
    reader_threads        host threads decoding frames (the Huffman stage of endo_jpeg_decode_crop releases the GIL; one FrameDecoder
                          per thread): 2.2 ms per 1920x1080 frame and thread, so 4 threads put a batch of 8 pairs together in ~9 ms
    prefetch              batches assembled ahead of the consumer by a producer thread on a side stream (the reference's DataLoader
                          workers); 0 = assemble in the caller's thread, on its stream
    """

    def __init__(self, folder_list, adjacent_range, batch_size, downsampling=4.0, network_downsampling=64, visible_interval=30,
                 precompute_path=None, image_file_names=None, num_iter=None, shuffle=True, rgb_mode="rgb", suggested_h=None,
                 suggested_w=None, device="cuda", seed=None, inlier_percentage=None, reader_threads=4, prefetch=1, is_hsv=False):
        assert len(adjacent_range) == 2
        self.folders = [str(f) for f in folder_list]
        self.adjacent_range = list(adjacent_range)
        self.batch_size = int(batch_size)
        self.downsampling = float(downsampling)
        self.rgb_mode = rgb_mode
        self.is_hsv = bool(is_hsv)          # train.py --use_hsv_colorspace: frames enter the network as cv2.COLOR_BGR2HSV_FULL values (dataset.py:439-442)
        self.shuffle = shuffle
        self.device = torch.device(device)
        self.rng = random.Random(seed)
        if image_file_names is None:
            image_file_names = []
            for folder in self.folders:
                image_file_names += [os.path.join(folder, n) for n in sorted(os.listdir(folder)) if n.startswith("0") and n.endswith(".jpg")]
        self.image_file_names = [str(n) for n in image_file_names]
        if not self.image_file_names:
            raise ValueError("no frames found below %s" % self.folders)
        self.num_iter = len(self.image_file_names) if num_iter is None else int(num_iter)
        stored = read_precompute_file(precompute_path) if precompute_path is not None else None
        self.sequences = {}
        for folder in self.folders:
            if stored is not None:
                seq = {k: _by_folder_name(stored[k], folder) for k in ("crop_positions", "visible_view_indexes", "point_cloud", "intrinsic_matrix",
                                                                       "mask_boundary", "view_indexes_per_point", "extrinsics", "projection",
                                                                       "clean_point_list", "estimated_scale")}
            else:
                # inlier_percentage (train.py: 0.99) runs the contaminated-point filter on the folder's frames; None = no filter
                seq = reader.load_sequence(folder, self.downsampling, network_downsampling, visible_interval, suggested_h, suggested_w,
                                           inlier_percentage)
                if seq["clean_point_list"] is None:
                    seq["clean_point_list"] = []
            seq["scatter"] = scatter.SequenceScatter(
                seq["point_cloud"], seq["mask_boundary"], seq["view_indexes_per_point"], seq["clean_point_list"], seq["visible_view_indexes"],
                device=self.device, extrinsics=np.stack([np.asarray(m) for m in seq["extrinsics"]]),
                projections=np.stack([np.asarray(m) for m in seq["projection"]]), intrinsic_matrix=seq["intrinsic_matrix"],
                estimated_scale=seq["estimated_scale"])
            self.sequences[folder] = seq
        self.reader_threads = max(1, int(reader_threads))
        self.prefetch = max(0, int(prefetch))
        self._pool = ThreadPoolExecutor(self.reader_threads) if self.reader_threads > 1 else None
        self._local = threading.local()
        self._side = torch.cuda.Stream(device=self.device) if self.prefetch else None
        # the sequences' resident tensors (SequenceScatter) were produced by asynchronous kernels on this thread's stream; the
        # producer thread reads them from the side stream
        if torch.cuda.is_available():
            torch.cuda.current_stream(self.device).synchronize()

    def _decode_into(self, path, window, dst, stream):
        """One frame into dst (3, H, W) on `stream`; runs on a reader thread with its own FrameDecoder."""
        decoder = getattr(self._local, "decoder", None)
        if decoder is None:
            decoder = self._local.decoder = reader.FrameDecoder(device=self.device, slots=4)
        with open(path, "rb") as f:
            data = f.read()
        with torch.cuda.device(self.device), torch.cuda.stream(stream):
            if self.is_hsv:
                scratch = torch.empty((dst.shape[1], dst.shape[2], 3), dtype=torch.uint8, device=self.device)
                decoder.decode(data, window[0], window[1], window[2], window[3], self.downsampling, "bgr", out_u8=scratch)
                reader.hsv_full(scratch, blue_index=0, out_f32=dst)
                scratch.record_stream(stream)
            else:
                decoder.decode(data, window[0], window[1], window[2], window[3], self.downsampling, self.rgb_mode, out_f32=dst)

    def __len__(self):
        return (self.num_iter + self.batch_size - 1) // self.batch_size

    def _draw(self, idx):
        """One sample: (folder, pos, increment) -- dataset.py:338-350."""
        name = self.image_file_names[idx % len(self.image_file_names)]
        folder = os.path.dirname(name)
        seq = self.sequences[folder]
        pos, increment = utils.generating_pos_and_increment(idx, seq["visible_view_indexes"], self.adjacent_range, rng=self.rng)
        return folder, pos, increment

    def _assemble(self, samples):
        """samples: list of (folder, pos, increment).  Returns (batch, per-sample validity)."""
        order, parts = [], []
        for folder in sorted(set(s[0] for s in samples)):
            rows = [i for i, s in enumerate(samples) if s[0] == folder]
            seq = self.sequences[folder]
            positions = [(samples[i][1], samples[i][2]) for i in rows]
            part = seq["scatter"].training_batch(positions)
            sh, eh, sw, ew = [int(v) for v in seq["crop_positions"]]
            views = seq["visible_view_indexes"]
            c1 = torch.empty((len(rows), 3, eh - sh, ew - sw), dtype=torch.float32, device=self.device)
            c2 = torch.empty_like(c1)
            stream = torch.cuda.current_stream(self.device)
            jobs = []
            for k, (pos, inc) in enumerate(positions):
                for dst, view in ((c1, views[pos]), (c2, views[pos + inc])):
                    jobs.append((os.path.join(folder, "%08d.jpg" % view), (sh, eh, sw, ew), dst[k], stream))
            if self._pool is None:
                for job in jobs:
                    self._decode_into(*job)
            else:
                for done in [self._pool.submit(self._decode_into, *job) for job in jobs]:
                    done.result()
            part["colors_1"], part["colors_2"] = c1, c2
            order += rows
            parts.append(part)
        batch = {k: torch.cat([p[k] for p in parts], dim=0) for k in parts[0]}
        if order != sorted(order):
            inverse = torch.tensor(np.argsort(order), device=self.device)
            batch = {k: v.index_select(0, inverse) for k, v in batch.items()}
        valid = (batch["sparse_depth_masks_1"].sum(dim=(1, 2, 3)) != 0) & (batch["sparse_depth_masks_2"].sum(dim=(1, 2, 3)) != 0)
        return batch, valid.tolist()

    def __iter__(self):
        if not self.prefetch:
            yield from self._batches()
            return
        # producer thread: assembles on the side stream, hands over (batch, event); the consumer's stream waits for the event and
        # the tensors are marked as used by it, so the caching allocator does not hand their memory back to the side stream early
        handover = queue.Queue(maxsize=self.prefetch)
        stop = threading.Event()

        def hand_over(item):
            """False when the consumer has left (the queue may be full and will never drain)."""
            while not stop.is_set():
                try:
                    handover.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def produce():
            try:
                with torch.cuda.device(self.device), torch.cuda.stream(self._side):
                    for batch in self._batches():
                        ready = torch.cuda.Event()
                        ready.record(self._side)
                        if not hand_over((batch, ready)):
                            return
                hand_over(None)
            except BaseException as exc:          # noqa: BLE001 -- re-raised in the consumer
                hand_over(exc)

        worker = threading.Thread(target=produce, daemon=True)
        worker.start()
        try:
            while True:
                item = handover.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                batch, ready = item
                current = torch.cuda.current_stream(self.device)
                current.wait_event(ready)
                for value in batch.values():
                    value.record_stream(current)
                yield batch
        finally:
            stop.set()
            worker.join(timeout=5.0)

    def _batches(self):
        indices = list(range(self.num_iter))
        if self.shuffle:
            self.rng.shuffle(indices)
        for start in range(0, len(indices), self.batch_size):
            samples = [self._draw(idx) for idx in indices[start:start + self.batch_size]]
            batch, valid = self._assemble(samples)
            attempts = 0
            while not all(valid):          # dataset.py:372-376: a pair without sparse points is replaced by a random one
                attempts += 1
                if attempts > 20:
                    raise RuntimeError("could not draw a pair with sparse points after 20 attempts")
                samples = [s if ok else self._draw(self.rng.randint(0, len(self.image_file_names) - 1)) for s, ok in zip(samples, valid)]
                batch, valid = self._assemble(samples)
            yield batch
