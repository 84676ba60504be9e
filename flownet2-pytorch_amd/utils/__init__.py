"""Data formats on either side of the hot path (the Middlebury .flo writer / reader)."""
