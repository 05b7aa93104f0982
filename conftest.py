import os


def pytest_xdist_auto_num_workers(config):
    """`-n auto` of pytest.ini: no worker processes on a GPU box (the -m gpu tests share one device and one model file in /tmp),
    otherwise one per core up to eight."""
    if os.path.exists("/dev/kfd"):
        return 0
    return max(1, min(8, os.cpu_count() or 1))
