"""rec_now_amd.util -- MI355X-native counterparts of rec_now/util (same module and symbol names)."""
