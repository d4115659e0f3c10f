"""python standins/run_script.py <script.py> [args...]: run a script written
against the reference (its examples/*.py) with this build on the path --
`paccmann_chemistry` resolves to the MI355X implementation, `pytoda` and
`torch.utils.tensorboard` to the stand-ins of this directory."""
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    if len(sys.argv) < 2:
        sys.exit(__doc__)
    for path in (HERE, ROOT):
        if path not in sys.path:
            sys.path.insert(0, path)
    import tensorboard_stub
    tensorboard_stub.install()
    script = sys.argv[1]
    sys.argv = sys.argv[1:]
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
