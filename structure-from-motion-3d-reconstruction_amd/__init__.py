"""sfmx — MI355X-native SfM hot path (KLT + RANSAC + local-BA build) behind a C ABI."""
