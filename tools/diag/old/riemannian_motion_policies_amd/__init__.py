"""MI355X-native RMP2 evaluation-and-pullback engine.

Public surface (same names as the reference's top-level modules):
    rmp.RmpCore, rmp.TargetPolicy, rmp.JointLimitAvoidance, rmp.ConfigurationSpaceBiasing,
    rmp2.TargetAttractor, rmp2.JointVelocityCap, rmp2.JointDamping, rmp2.ObstacleAvoidance,
    rmp2.CSpaceBiasing, taskmap.*, kinematics.UrdfForwardKinematic, data_management.Datamanager
plus the fleet-level API: engine.Engine (device tensors in / out), fleet.Fleet (multi-GPU).
"""
__version__ = "0.1.0"
