'''Clock / power sampler for bench.py -- a CHILD process, standard library + amdsmi only, never HIP.

Boxes of the gpurun pool differ by a few percent on identical binaries (the MFMA loops are
power-capped); this makes the difference a number.  Samples (time.time(), gfx clock MHz, socket
power W, [mem clock MHz]) every --period seconds (default 0.1) until stdin closes, then prints one JSON object
{"source": ..., "samples": [[t, sclk, power, mclk], ...]}.  bench.py starts it before its first GPU
call and averages the samples that fall inside the timed region.

Sources, first that works: (1) amdsmi (amdsmi_get_gpu_metrics_info: current_gfxclk,
average_socket_power / current_socket_power); (2) sysfs: /sys/class/drm/card*/device/pp_dpm_sclk
(the starred level) and hwmon power1_average / power1_input (uW).

    python tools/devmon.py --probe      # print what is readable and one sample, then exit
'''
import glob
import json
import os
import sys
import time


def _num(v):
    try:
        if v in (None, 'N/A'):
            return None
        f = float(v)
        return f if f == f and f < 65535 else None
    except (TypeError, ValueError):
        return None


class SmiSource():
    name = 'amdsmi'

    def __init__(self, index=0):
        import amdsmi
        self.smi = amdsmi
        amdsmi.amdsmi_init()
        handles = amdsmi.amdsmi_get_processor_handles()
        self.h = handles[physical_index(index, len(handles))]
        self.read()                                        # raises if the call is not permitted
        try:
            self.bdf = str(amdsmi.amdsmi_get_gpu_device_bdf(self.h)).lower()
        except Exception:                                  # noqa: BLE001
            self.bdf = None

    def read(self):
        m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
        sclk = _num(m.get('current_gfxclk'))
        if sclk is None:
            clks = [c for c in (_num(x) for x in (m.get('current_gfxclks') or [])) if c]
            sclk = sum(clks) / len(clks) if clks else None
        power = _num(m.get('current_socket_power'))
        if power is None:
            power = _num(m.get('average_socket_power'))
        return sclk, power, _num(m.get('current_uclk'))


class SysfsSource():
    name = 'sysfs'

    def __init__(self, index=0):
        cards = []
        for d in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
            try:
                if open(os.path.join(d, 'vendor')).read().strip() == '0x1002' and \
                        os.path.exists(os.path.join(d, 'pp_dpm_sclk')):
                    cards.append(d)
            except OSError:
                pass
        self.dev = cards[physical_index(index, len(cards))]
        self.bdf = os.path.basename(os.path.realpath(self.dev)).lower()
        hw = glob.glob(os.path.join(self.dev, 'hwmon', 'hwmon*'))
        self.power = None
        for h in hw:
            for n in ('power1_average', 'power1_input'):
                if os.path.exists(os.path.join(h, n)):
                    self.power = os.path.join(h, n)
                    break
        self.read()

    @staticmethod
    def _starred(path):
        for line in open(path).read().splitlines():
            if line.rstrip().endswith('*'):
                return _num(line.split(':')[1].strip().rstrip('*').strip().lower().replace('mhz', ''))
        return None

    def read(self):
        sclk = self._starred(os.path.join(self.dev, 'pp_dpm_sclk'))
        power = _num(open(self.power).read()) if self.power else None
        mclk = None
        try:
            mclk = self._starred(os.path.join(self.dev, 'pp_dpm_mclk'))
        except OSError:
            pass
        return sclk, (power / 1e6 if power is not None else None), mclk


def physical_index(index, n_all):
    '''amdsmi and sysfs enumerate EVERY GPU of the machine; the bench process numbers the ones its runtime shows it.  Translate
    the runtime's ordinal through ROCR_VISIBLE_DEVICES (applied first) and HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES (an index
    into what ROCR left).  Entries that are not plain integers (UUIDs) cannot be mapped here: the ordinal is used as is and the
    caller compares the sampled device's PCI address with the one the runtime reports (bench.py `clock_device_matches`).'''
    phys = list(range(n_all))
    for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES' if 'HIP_VISIBLE_DEVICES' in os.environ else 'CUDA_VISIBLE_DEVICES'):
        val = os.environ.get(var)
        if val is None or val.strip() == '':
            continue
        try:
            sel = [int(v) for v in val.split(',') if v.strip() != '']
            phys = [phys[i] for i in sel if 0 <= i < len(phys)]
        except ValueError:
            return index
    return phys[index] if 0 <= index < len(phys) else index


def open_source(index=0):
    errs = []
    for cls in (SmiSource, SysfsSource):
        try:
            return cls(index), errs
        except Exception as ex:           # noqa: BLE001 -- any failure means: try the next source
            errs.append(f'{cls.name}: {type(ex).__name__}: {ex}')
    return None, errs


def main():
    index = int(os.environ.get('FD_DEVMON_INDEX', '0'))
    period = 0.1
    if '--period' in sys.argv:
        period = float(sys.argv[sys.argv.index('--period') + 1])
    src, errs = open_source(index)
    if '--probe' in sys.argv:
        print(json.dumps({'source': src.name if src else None, 'errors': errs,
                          'sample': src.read() if src else None}))
        return
    if src is None:
        sys.stdin.read()
        print(json.dumps({'source': None, 'errors': errs, 'samples': []}))
        return
    import select
    samples = []
    while True:
        try:
            samples.append((time.time(),) + tuple(src.read()))
        except Exception as ex:           # noqa: BLE001
            errs.append(str(ex))
            break
        r, _, _ = select.select([sys.stdin], [], [], period)
        if r and not sys.stdin.readline():
            break
    print(json.dumps({'source': src.name, 'errors': errs[:3], 'bdf': getattr(src, 'bdf', None), 'samples': samples}))


if __name__ == '__main__':
    main()
