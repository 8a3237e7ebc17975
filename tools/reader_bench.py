"""Colour-frame reader timing: .jpg bytes -> cropped, downsampled frame in HBM.
usage: python tools/reader_bench.py [repeats]     (run on the GPU box; prints one JSON line)
Per 1920x1080 4:2:0 frame of the example sequence: the host stage (parse + Huffman decode into pinned memory), the device
stage (copy + inverse DCT + resize/crop kernels, HIP events on the launch stream), frames/s with both pipelined over the
decoder's slots, and the CPU path the reference runs (libjpeg-turbo decode + the oracle's cv2.resize restatement)."""
import ctypes
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ea = importlib.import_module("endoscopydepthestimation-pytorch_amd")
reader = importlib.import_module("endoscopydepthestimation-pytorch_amd.reader")
from oracle import reader as oreader  # noqa: E402  (the CPU baseline only)


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seq = os.path.join(ROOT, "tests", "golden", "example_sequence", "bag_1", "_start_004259_end_004629_stride_25_segment_13")
    raws = [open(os.path.join(seq, n), "rb").read() for n in ("00004584.jpg", "00004594.jpg")]
    crop = (11, 267, 88, 408)
    lib = ea._lib.load()
    # host stage alone
    buf = np.frombuffer(raws[0], np.uint8)
    info = np.zeros(16, np.int32)
    lib.endo_jpeg_info(ctypes.c_void_p(buf.ctypes.data), len(raws[0]), ctypes.c_void_p(info.ctypes.data))
    blocks = np.zeros((int(info[13]), 64), np.int16)
    quant = np.zeros((3, 64), np.uint16)
    t0 = time.perf_counter()
    for _ in range(reps):
        lib.endo_jpeg_entropy_decode(ctypes.c_void_p(buf.ctypes.data), len(raws[0]), ctypes.c_void_p(blocks.ctypes.data), int(info[13]),
                                     ctypes.c_void_p(quant.ctypes.data))
    host_ms = (time.perf_counter() - t0) / reps * 1e3
    decoder = reader.FrameDecoder(slots=4)
    out = torch.empty((3, 256, 320), dtype=torch.float32, device="cuda")
    for _ in range(3):
        decoder.decode(raws[0], *crop, 4.0, "rgb", out_f32=out)
    torch.cuda.synchronize()
    # device stage: events around one call (the host stage runs before the first launch of the call)
    dev = []
    for _ in range(10):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        decoder.decode(raws[0], *crop, 4.0, "rgb", out_f32=out)
        b.record()
        torch.cuda.synchronize()
        dev.append(a.elapsed_time(b))
    t0 = time.perf_counter()
    for i in range(reps):
        decoder.decode(raws[i & 1], *crop, 4.0, "rgb", out_f32=out)
    torch.cuda.synchronize()
    pipelined_ms = (time.perf_counter() - t0) / reps * 1e3
    # reader threads, one FrameDecoder each (the Huffman stage releases the GIL)
    import threading
    threaded = {}
    for nthreads in (2, 4):
        def work():
            dec = reader.FrameDecoder(slots=4)
            dst = torch.empty((3, 256, 320), dtype=torch.float32, device="cuda")
            for i in range(reps):
                dec.decode(raws[i & 1], *crop, 4.0, "rgb", out_f32=dst)
            torch.cuda.synchronize()
        ts = [threading.Thread(target=work) for _ in range(nthreads)]
        t0 = time.perf_counter()
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        threaded[nthreads] = round(nthreads * reps / (time.perf_counter() - t0), 1)
    t0 = time.perf_counter()
    cpu_reps = max(reps // 5, 3)
    for i in range(cpu_reps):
        rgb = oreader.decode_jpeg_pil(raws[i & 1])
        small = oreader.resize_linear(rgb, 4.0)[crop[0]:crop[1], crop[2]:crop[3]]
        _ = ((small.astype(np.float32) - np.float32(127.5)) * np.reciprocal(np.float32(127.5))).transpose(2, 0, 1).copy()
    cpu_ms = (time.perf_counter() - t0) / cpu_reps * 1e3
    t0 = time.perf_counter()
    for i in range(cpu_reps):
        oreader.decode_jpeg_pil(raws[i & 1])
    lib_ms = (time.perf_counter() - t0) / cpu_reps * 1e3
    print(json.dumps({"frame": "1920x1080 4:2:0 -> 256x320 crop of the 1/4 image", "host_huffman_ms": round(host_ms, 3),
                      "device_stage_ms_incl_host_call": round(float(np.median(dev)), 3), "pipelined_ms_per_frame": round(pipelined_ms, 3),
                      "frames_per_s_one_thread": round(1e3 / pipelined_ms, 1), "frames_per_s_by_threads": threaded, "cpu_libjpeg_turbo_decode_ms": round(lib_ms, 3),
                      "cpu_decode_resize_normalise_ms (oracle: Pillow + numpy)": round(cpu_ms, 3), "file_bytes": len(raws[0])}))


if __name__ == "__main__":
    main()
