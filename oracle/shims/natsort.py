# Test-infrastructure shim (dataset loaders only; never on the hot path).
def natsorted(x):
    return sorted(x)
