"""rec_now_amd.layers -- MI355X-native counterparts of rec_now/layers (same module and symbol names)."""
