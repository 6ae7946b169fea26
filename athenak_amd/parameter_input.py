"""ParameterInput: reader for the reference's athinput decks.

Mirrors src/parameter_input.hpp:67-127 / parameter_input.cpp:155-209,369-409,508-552:
`<block>` headers, `name = value # comment`, `<par_end>` stops parsing, `#` comment lines,
command-line overrides `block/name=value` that may only REPLACE existing parameters
(a missing block or name is fatal), booleans accept 0/1/true/false (case-insensitive).
"""


class ParameterInputError(RuntimeError):
    pass


class ParameterInput:
    def __init__(self, text=None, filename=None):
        self.blocks = {}          # name -> dict (insertion ordered)
        if filename is not None:
            with open(filename) as f:
                text = f.read()
        if text is not None:
            self.LoadFromString(text)

    def LoadFromString(self, text):
        block = None
        for raw in text.splitlines():
            line = raw.replace("\t", " ").strip()
            if not line or line.startswith("#"):
                continue
            if line.startswith("<"):
                if ">" not in line:                      # parameter_input.cpp:176-181
                    raise ParameterInputError("### FATAL ERROR Block name '%s' not properly ended" % line[1:])
                name = line[1:line.index(">")].strip()
                if name == "par_end":
                    break
                # a repeated <block> continues the first one and a repeated name replaces the value:
                # FindOrAddBlock / AddParameter, parameter_input.cpp:266-282,328-352
                block = self.blocks.setdefault(name, {})
                continue
            if block is None:
                raise ParameterInputError("### FATAL ERROR parameter outside of a <block>: " + raw)
            if "=" not in line:
                raise ParameterInputError("### FATAL ERROR no '=' in line: " + raw)
            k, v = line.split("=", 1)
            v = v.split("#", 1)[0].strip()
            block[k.strip()] = v

    def ModifyFromCmdline(self, args):
        """parameter_input.cpp:369-409: override existing parameters only."""
        for a in args:
            if "/" not in a or "=" not in a:
                raise ParameterInputError("### FATAL ERROR cannot parse override: " + a)
            blk, rest = a.split("/", 1)
            name, val = rest.split("=", 1)
            if blk not in self.blocks:
                raise ParameterInputError("### FATAL ERROR block <%s> not found" % blk)
            if name not in self.blocks[blk]:
                raise ParameterInputError("### FATAL ERROR parameter %s/%s not found" % (blk, name))
            self.blocks[blk][name] = val

    def DoesBlockExist(self, blk):
        return blk in self.blocks

    def DoesParameterExist(self, blk, name):
        return blk in self.blocks and name in self.blocks[blk]

    def _get(self, blk, name):
        if not self.DoesParameterExist(blk, name):
            raise ParameterInputError("### FATAL ERROR parameter %s/%s does not exist" % (blk, name))
        return self.blocks[blk][name]

    def GetString(self, blk, name):
        return self._get(blk, name)

    def GetInteger(self, blk, name):
        return int(self._get(blk, name))

    def GetReal(self, blk, name):
        return float(self._get(blk, name))

    def GetBoolean(self, blk, name):
        v = self._get(blk, name).lower()
        if v in ("1", "true"):
            return True
        if v in ("0", "false"):
            return False
        raise ParameterInputError("### FATAL ERROR bad boolean %s/%s=%s" % (blk, name, v))

    def _get_or_add(self, blk, name, default, conv):
        if self.DoesParameterExist(blk, name):
            return conv(blk, name)
        self.blocks.setdefault(blk, {})[name] = str(default)
        return default

    def GetOrAddString(self, blk, name, default):
        return self._get_or_add(blk, name, default, self.GetString)

    def GetOrAddInteger(self, blk, name, default):
        return self._get_or_add(blk, name, default, self.GetInteger)

    def GetOrAddReal(self, blk, name, default):
        return self._get_or_add(blk, name, default, self.GetReal)

    def GetOrAddBoolean(self, blk, name, default):
        return self._get_or_add(blk, name, default, self.GetBoolean)

    def SetReal(self, blk, name, val):
        """src/parameter_input.cpp:722-731: the value is stored as `stringstream << Real`, i.e.
        with 6 significant digits ("%g").  This is observable: the linear-wave generator passes
        its rescaled time limit through here (2.999999999997 -> 3, 1.4999999787 -> 1.5)."""
        self.blocks.setdefault(blk, {})[name] = "%g" % float(val)

    def Dump(self):
        """deck text of the current state (ParameterDump of the reference, src/main.cpp:380)"""
        out = []
        for b, d in self.blocks.items():
            out.append("<%s>" % b)
            out += ["%s = %s" % (k, v) for k, v in d.items()]
        return "\n".join(out) + "\n"

    def SetInteger(self, blk, name, val):
        self.blocks.setdefault(blk, {})[name] = str(int(val))

    def ParameterDump(self):
        """src/parameter_input.cpp:771-795: the text that heads bin files (and restarts); names
        and values padded to the longest of their block"""
        out = ["#------------------------- PAR_DUMP -------------------------"]
        for b, d in self.blocks.items():
            out.append("<%s>" % b)
            ln = max([len(k) for k in d] + [0])
            lv = max([len(str(v)) for v in d.values()] + [0])
            for k, v in d.items():
                out.append("%s= %s" % (k.ljust(ln + 1), str(v).ljust(lv + 1)))
        out.append("#------------------------- PAR_DUMP -------------------------")
        out.append("<par_end>")
        return "\n".join(out) + "\n"

    def SetString(self, blk, name, val):
        self.blocks.setdefault(blk, {})[name] = str(val)
