"""`python -m src.scripts.train_sae --config <json>` -- the reference's entry point name
(README.md:52,61), forwarding to the MI355X engine's host (freud_amd/train_sae.py)."""
from freud_amd.train_sae import main, train  # noqa: F401

if __name__ == "__main__":
    main()
