# Test-infrastructure shim: the reference imports open3d only for visualisation.
