"""Drop-in entry point: `python train_nerv_all.py <reference flags>` (scripts/regression/**.sh of the reference run unchanged)."""
from boosting_nerv_amd.train_nerv_all import *  # noqa: F401,F403
from boosting_nerv_amd.train_nerv_all import main

if __name__ == '__main__':
    main()
