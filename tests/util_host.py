"""Host identification used to decide whether bit-equality with a fixture can be demanded."""


def cpu_vendor():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("vendor_id"):
                return line.split(":")[1].strip()
    except OSError:
        pass
    return "unknown"
