"""Host-side mirror of the reference's ``social_gym`` package for the crowd-step path.

When gymnasium is importable the env is registered under the reference's id ``SocialGym-v0``
(reference: social_gym/__init__.py:3-6)."""
try:  # pragma: no cover - gymnasium is not part of the build image
    from gymnasium.envs.registration import register

    register(id="SocialGym-v0", entry_point="social_navigation_pyenvs_amd.social_gym.social_nav_gym:SocialNavGym")
except Exception:
    pass
