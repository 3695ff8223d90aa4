"""`python -m v2a_hip.launch <script.py> [script arguments]`: run an UNCHANGED script of the user's video-to-action checkout (e.g. the
reference's scripts/train_libero_dp.py) on the MI355X-native hot path -- installs the import overlay (v2a_hip/overlay.py), then executes
the script as `__main__` with `sys.argv` and `sys.path[0]` as `python <script.py>` would have set them."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    import v2a_hip                                   # raises if libv2a_hip.so is missing: no silent fallback
    from v2a_hip import overlay
    overlay.install()
    script = argv[0]
    sys.argv = argv
    sys.path[0] = os.path.dirname(os.path.abspath(script))
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
