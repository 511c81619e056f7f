"""Socket power / shader clock of the GPU THIS process computes on, polled on a thread (round 5: tools/energy_layers.py,
tools/energy_tune.py).  In-process through amdsmi when the handle can be matched to torch's device by PCI bus id (20 ms period),
else `rocm-smi` as a subprocess (250 ms).  Never the hwmon files of "card0": on a multi-GPU host that is somebody else's board."""
import re
import subprocess
import threading
import time


class PowerMeter(object):
    def __init__(self, device_index=0):
        self.h = None
        self.smi = None
        self.source = 'rocm-smi (subprocess)'
        self.period = 0.25
        try:
            import torch
            import amdsmi
            pr = torch.cuda.get_device_properties(device_index)
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            want = '%04x:%02x:%02x' % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            match = [h for h in hs if str(amdsmi.amdsmi_get_gpu_device_bdf(h)).lower().startswith(want)]
            if len(match) == 1 or len(hs) == 1:
                self.h = match[0] if len(match) == 1 else hs[0]
                self.smi = amdsmi
                self.source = 'amdsmi (%s)' % amdsmi.amdsmi_get_gpu_device_bdf(self.h)
                self.period = 0.02
                if self.read() is None:
                    self.h = None
        except Exception as exc:      # no amdsmi / no match: the subprocess form
            self.h = None
            self.source = 'rocm-smi (subprocess; amdsmi: %s)' % type(exc).__name__
        if self.h is None:
            self.period = 0.25
        self.samples, self._stop, self._th = [], True, None

    def read(self):
        """-> (watts, shader MHz) or None."""
        if self.h is not None:
            try:
                p = self.smi.amdsmi_get_power_info(self.h)
                w = p.get('current_socket_power')
                if not isinstance(w, (int, float)) or w <= 0:
                    w = p.get('average_socket_power')
                c = self.smi.amdsmi_get_clock_info(self.h, self.smi.AmdSmiClkType.GFX).get('clk', -1)
                return (float(w), float(c)) if isinstance(w, (int, float)) and w > 0 else None
            except Exception:
                return None
        try:
            o = subprocess.run(['rocm-smi', '--showpower', '--showclocks'], capture_output=True, text=True, timeout=5).stdout
            pw = re.search(r'Power \(W\): ([0-9.]+)', o)
            ck = re.search(r'sclk clock level: \d+: \((\d+)Mhz\)', o)
            return (float(pw.group(1)), float(ck.group(1)) if ck else -1.0) if pw else None
        except Exception:
            return None

    def start(self):
        self.samples, self._stop = [], False

        def poll():
            while not self._stop:
                r = self.read()
                if r is not None:
                    self.samples.append((time.perf_counter(),) + r)
                time.sleep(self.period)
        self._th = threading.Thread(target=poll, daemon=True)
        self._th.start()

    def stop(self, t_from=0.0):
        """-> (mean watts, mean MHz, samples) over the samples taken at or after perf_counter() == t_from."""
        self._stop = True
        self._th.join(timeout=10)
        sm = [s for s in self.samples if s[0] >= t_from]
        if not sm:
            return None, None, 0
        return sum(s[1] for s in sm) / len(sm), sum(s[2] for s in sm) / len(sm), len(sm)
