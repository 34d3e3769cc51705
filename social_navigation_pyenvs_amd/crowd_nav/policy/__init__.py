"""Device-side array helpers of the CrowdNav policies (look-ahead); the policies themselves are user code."""
