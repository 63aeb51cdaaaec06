"""Stage times of rtx_scene_create (RTX_BUILD_TIMING) for a named scene, cold and warm: gpu_buildtime.py [scene]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RTX_BUILD_TIMING"] = "1"
pkg = importlib.import_module("rust-raytracer_amd")
name = sys.argv[1] if len(sys.argv) > 1 else "sponza"
named = pkg.host.NamedScene(name)
for i in range(3):
    t0 = time.time()
    sc = pkg.Scene(named, device=0)
    print("---- build %d: %.1f ms" % (i, (time.time() - t0) * 1e3), file=sys.stderr)
    sc.close()
