"""CrowdNav state / action records exchanged at the Gym boundary."""
