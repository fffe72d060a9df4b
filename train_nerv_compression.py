"""Drop-in entry point: the reference's train_nerv_compression.py CLI, served by the MI355X build."""
from boosting_nerv_amd.train_nerv_compression import *  # noqa: F401,F403
from boosting_nerv_amd.train_nerv_compression import main

if __name__ == "__main__":
    main()
