"""Run an unmodified reference driver with the HIP overlay in front of its own modules:

    cd <reference checkout>
    python -m geometrics_amd.run GEOMetrics.py [driver args...]

`python GEOMetrics.py` puts the script's directory FIRST on sys.path, so a PYTHONPATH entry can
never shadow the reference's `chamfer_distance/`, `tri_distance/`, `layers.py`, `utils.py`.
This launcher builds sys.path as [overlay, repo, script dir, ...] and then executes the
driver as __main__.
"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit(__doc__)
    script = os.path.abspath(argv[0])
    script_dir = os.path.dirname(script)
    front = [os.path.join(ROOT, "overlay"), ROOT, script_dir]
    sys.path[:] = front + [p for p in sys.path if os.path.abspath(p or os.getcwd()) not in front]
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()
