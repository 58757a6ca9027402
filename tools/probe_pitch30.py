import importlib, sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
pano = synth.synth_pano(16384, 8192, 1000, "S")
ctx = nat.Context(0)
Y = list(range(0, 360, 5))
for p in (30, 60):
    job = nat.Job(ctx, 16384, 8192, 1, Y, [p], 60, 4096, 4096)
    job.set_pano(0, pano)
    for _ in range(30):
        job.run()
    job.close()
