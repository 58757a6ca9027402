import importlib, sys, os
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
pano = synth.synth_pano(8192, 4096, 1000, "S")
ctx = nat.Context(0)
def run(yaws, pitches, label, n=200):
    job = nat.Job(ctx, 8192, 4096, 1, yaws, pitches, 90, 1920, 1080)
    job.set_pano(0, pano); job.time_launches(False)
    for _ in range(60): job.run()
    ctx.mark(0)
    for _ in range(n): job.run()
    ctx.mark(1)
    ms = ctx.marked_ms() / n
    print("%-44s %8.1f us" % (label, ms*1e3), flush=True)
    job.close()
copy=[0,45,90,135,180,225,270,315,0,45,90,135]
frac=list(range(0,360,30))
for p in (90,45,30):
    run(copy,[p],"pitch %d copy yaws"%p)
    run(frac,[p],"pitch %d 30-degree yaws"%p)
