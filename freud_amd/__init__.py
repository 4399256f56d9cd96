"""freud_amd -- MI355X-native SAE training engine, drop-in for ksadov/FREUD's train_sae path."""
__version__ = "0.1.0"
