"""Worker process of tests/test_gpu_reference_kernel.py: renders one of tests/test_oracle.py's CASES with whatever build of the library the environment
variable DSRT_LIB names (the test points it at oracle/_ref/libdsrt_hip_devlibm.so) and writes the rgb8 image, top row first, to a file.  A process
of its own because two builds of one library cannot share a process."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    name, out = sys.argv[1], sys.argv[2]
    import dsrt_amd as d
    from conftest import load_world
    from test_oracle import CASES, SUN
    world, cam_args, spp = CASES[name]
    hs = load_world(d, world)
    W, H, depth = cam_args[3], cam_args[4], cam_args[5]
    cam = d.camera_look_at(cam_args[0], cam_args[1], cam_args[2], W, H, spp, depth)
    ctx = d.Context(0)
    ctx.upload(hs.view(cam, SUN))
    rgb, _, _ = ctx.render_to_host(d.make_desc(W, H, spp, depth))
    with open(out, "wb") as f:
        f.write(rgb.tobytes())
    ctx.close()


if __name__ == "__main__":
    main()
