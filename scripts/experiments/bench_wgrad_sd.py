import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import bench_wgrad as bw
for cfg in [(320, 320, 64), (640, 320, 64), (960, 320, 64), (640, 640, 32), (1280, 640, 32), (1920, 640, 32), (1280, 1280, 16), (2560, 1280, 16), (1280, 1280, 8), (2560, 1280, 8)]:
    bw.bench(32, cfg[0], cfg[1], cfg[2], cfg[2], affine=False)
