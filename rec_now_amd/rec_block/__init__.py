"""rec_now_amd.rec_block -- MI355X-native counterparts of rec_now/rec_block (same module and symbol names)."""
