import importlib, sys, os
sys.path.insert(0,'/root/repo')
pkg = importlib.import_module("360-to-planer-images_amd"); nat = pkg._native
synth = importlib.import_module("360-to-planer-images_amd.synth")
pano = synth.synth_pano(8192, 4096, 1000, "S")
ctx = nat.Context(0)
def run(yaws, pitches, label, n=300):
    job = nat.Job(ctx, 8192, 4096, 1, yaws, pitches, 90, 1920, 1080)
    job.set_pano(0, pano); job.time_launches(False)
    for _ in range(100): job.run()
    ctx.mark(0)
    for _ in range(n): job.run()
    ctx.mark(1)
    ms = ctx.marked_ms() / n
    print("%-40s %8.1f us  %6.2f us/yaw" % (label, ms*1e3, ms*1e3/len(yaws)), flush=True)
    job.close()
for ny in (24, 48):
    yaws=[(i*360)//ny for i in range(ny)]
    for ppb in (6, 12, 24, 48):
        if ppb>ny: continue
        os.environ["P2P_PAIRS_PER_BLOCK"]=str(ppb); nat.reload_options()
        run(yaws,[60,90,120],"%d yaws ppb %d"%(ny,ppb))
