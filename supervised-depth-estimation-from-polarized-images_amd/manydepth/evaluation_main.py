"""reference manydepth/evaluation_main.py:7-10."""
from manydepth.evaluation import Evaluation


def main():
    ev = Evaluation()
    ev.load_mono_model()
    ev.test()


if __name__ == "__main__":
    main()
