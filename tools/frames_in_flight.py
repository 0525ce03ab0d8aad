#!/usr/bin/env python3
"""Short launches back to back: images per second with one image in flight (render, wait, render ...) and with two (two contexts on
two streams, so that the drain of one launch -- its last long paths -- overlaps the start of the next).
usage: tools/frames_in_flight.py [spp=10] [frames=40] [width=1200] [height=800]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import raytracing_amd as R

spp = int(sys.argv[1]) if len(sys.argv) > 1 else 10
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 40
width = int(sys.argv[3]) if len(sys.argv) > 3 else 1200
height = int(sys.argv[4]) if len(sys.argv) > 4 else 800
dev = torch.device("cuda", 0)
for lists in (False, True):
    ctxs, outs, streams = [], [], []
    for k in range(2):
        rt = R.Renderer(0, camera_ray_lists=lists)
        rt.upload_scene(R.final_scene_table()); rt.set_camera(R.final_camera(width, height))
        ctxs.append(rt); outs.append(torch.zeros((height, width, 3), dtype=torch.int32, device=dev)); streams.append(torch.cuda.Stream())
    p = [rt.params(width, height, spp, seed=0) for rt in ctxs]
    def run(n_in_flight):
        for k in range(2):
            ctxs[k].render_device(p[k], d_irgb=outs[k].data_ptr(), stream=streams[k].cuda_stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for f in range(frames):
            k = f % n_in_flight
            ctxs[k].render_device(p[k], d_irgb=outs[k].data_ptr(), stream=streams[k].cuda_stream)
            if n_in_flight == 1:
                streams[k].synchronize()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / frames * 1e3
    one = min(run(1) for _ in range(3)); two = min(run(2) for _ in range(3))
    same = bool(torch.equal(outs[0], outs[1]))
    print(f"{width}x{height}x{spp}, camera-ray lists {'on' if lists else 'off'}: one image in flight {one:.3f} ms per image, two in flight {two:.3f} ms per image ({one / two:.2f}x), images equal: {same}", flush=True)
    for rt in ctxs:
        rt.close()
