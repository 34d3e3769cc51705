"""The CrowdNav data types exchanged at the Gym boundary (reference: crowd_nav/utils/state.py, action.py)."""
