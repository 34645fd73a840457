#!/usr/bin/env python3
"""gpurun_out/r06b (tools/collect_r06b.sh) -> profiles/r06_bench.json.txt, r06_bench_extras.json, r06_lsd.txt, r06_images_with_lines.txt."""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); O = os.path.join(R, "gpurun_out", "r06b") + "/"; P = os.path.join(R, "profiles") + "/"
rd = lambda f: open(O + f).read()
line = rd("bench_default.json").strip().splitlines()[-1]
d = json.loads(line); lg = d["legs"]
hdr = ("# python bench.py --gpus 1 --steps 20 --warmup 5 (the driver's command) on one MI355X, round 6 FINAL code (second session: tools/collect_r06b.sh).  stdout = this ONE line (%d bytes);\n"
       "# the full record of the same run (every leg, every note) is profiles/r06_bench_extras.json (bench_extras.json beside bench.py).  3072 streams per GPU; the hot path is the first\n"
       "# session's (1.35 - 1.40 M on the boxes of the pool), the LSD / LBD legs are this session's: %.1f k images/s at 8192 per launch, one image %.1f ms, images -> poses with key-lines\n"
       "# %.1f k pairs/s at 3072 streams, one stereo pair with key-lines %.1f ms from images to pose.\n") % (
    len(line), lg["lsd_images_per_s"] / 1e3, lg["lsd_one_image_ms"], lg["images_to_poses_with_lines_pairs_per_s"] / 1e3, lg["one_stereo_pair_with_lines_ms"])
open(P + "r06_bench.json.txt", "w").write(hdr + line + "\n")
open(P + "r06_bench_extras.json", "w").write(rd("bench_extras_default.json"))
parts = ["# LSD, round 6 FINAL code (tools/collect_r06b.sh).  (a) the many-waves form of small batches (lsd_grow_xcd_kernel: one XCD per image, 64 speculating waves): tools/lsd_probe.py,\n"
         "# one / two / eight KITTI-size images per call (lsd_scale 1.2: 672 k pixels), the committer's cycle counters, the speculating waves' and the dispatcher's;\n"
         "# round 5 (one workgroup of 16 waves): 22.8 / 24.7 / 26.8 ms per call.\n", rd("lsd_small_batches.txt"),
         "\n# (b) kernel stats of ONE image per call (rocprofv3 --kernel-trace --stats -- python tools/lsd_probe.py --batch 1 --iters 3)\n", rd("lsd_one_image_kernel_stats.txt"),
         "\n# (c) batches of 16 ... 128 images: the many-waves form (several images per XCD) against one wave per image (STVO_LSD_WAVES=0), ms per call incl. the host copies\n", rd("lsd_mid_batches.txt"),
         "\n# (d) 4096 images per launch, device-resident (bench.py's leg, tools/r06_lsd_leg.py 4096, under rocprofv3 --kernel-trace --stats): the leg's line, then per kernel\n"
         "# (8 launches of 4096 images + 6 of one image; round 5 per launch: growth 121 ms, gradient 22, rocPRIM sorts 22, key kernel 14, resize 7 = 188 ms; LBD 38 ms)\n", rd("lsd_batched_leg.json"), rd("lsd_batched_kernel_stats.txt"),
         "\n# (e) SQ counters of lsd_grow_kernel<true> (one wave per image) at 1024 images = one wave per SIMD (tools/r06_lsd_sq.sh; SQ_WAVE_CYCLES / SQ_ACTIVE_* count quad-cycles):\n"
         "# per image 9.3 M vector + 7.8 M scalar instructions, an instruction active 41 % of the wave's life, waiting 54 %\n", rd("lsd_one_wave_sq_counters.txt")]
open(P + "r06_lsd.txt", "w").write("".join(parts))
open(P + "r06_images_with_lines.txt", "w").write("# images -> poses with key-lines at 2048 streams (bench.py's leg alone: tools/r06_images_leg.py 2048 1 3 under rocprofv3 --kernel-trace --stats; tools/r06_images_prof.sh), round 6 FINAL code\n"
                                                 "# (bench.py runs the leg at 3072 streams: %.1f k pairs/s)\n" % (lg["images_to_poses_with_lines_pairs_per_s"] / 1e3) + rd("images_with_lines_leg.json") + rd("images_with_lines_kernel_stats.txt"))
print("profiles written:", d["value"], lg)
