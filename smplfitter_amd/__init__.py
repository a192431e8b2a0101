"""smplfitter_amd — MI355X-native implementation of smplfitter's BodyFitter.fit() hot path."""
