"""Agent records, obstacle geometry, the array seam and the motion-model manager of the crowd-step path."""
