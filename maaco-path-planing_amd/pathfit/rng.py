"""Per-agent keyed random streams.

The reference draws every random number from one global, never-seeded
``random`` / ``numpy.random`` stream shared by all agents in sequence
(MAACO.py:232,250,254,259,262; MPA.py:248-279,343-391;
pso.py:50-51,105,186-190; ga_solver.py:50-51,139,145-158).  Draw counts are
data dependent, so agent k's position in that stream depends on agent k-1
having finished: no agent-parallel engine can replay it.  The contract of this
engine is therefore *per agent-call*: agent ``a`` of iteration ``it`` in
domain ``dom`` owns the counter-based stream keyed ``(seed, dom, it, a)``.

``AgentRandom`` is a ``random.Random`` subclass that overrides only
``random()`` and ``getrandbits()``; CPython's own ``random.py`` then supplies
``choice`` / ``randint`` / ``uniform`` / ``normalvariate`` / ``sample`` exactly
as the reference would get them.  The HIP kernels (csrc/pf_rng.h) implement the
same generator and restate those CPython 3.10 derivations.
"""
import random as _random

_M = (1 << 64) - 1
GOLDEN = 0x9E3779B97F4A7C15

# stream domains
DOM_MAACO = 1       # MAACO ant walk            (agent = ant index)
DOM_MPA = 2         # MPA phase loop            (agent = predator index)
DOM_PSO = 3         # PSO velocity/position     (agent = particle index)
DOM_GA = 4          # GA crossover/mutation     (agent = child pair index)
DOM_MPA_FADS = 5    # MPA FADs loop             (agent = predator index)
DOM_INIT = 6        # population initialisation (agent = attempt index)
DOM_GA_SELECT = 7   # GA tournament selection   (agent = slot index)


def mix64(z):
    z &= _M
    z ^= z >> 30
    z = (z * 0xBF58476D1CE4E5B9) & _M
    z ^= z >> 27
    z = (z * 0x94D049BB133111EB) & _M
    z ^= z >> 31
    return z


def stream_key(seed, dom, it, agent):
    k = mix64(seed + GOLDEN * (dom + 1))
    k = mix64(k + 0xD1B54A32D192ED03 * (it + 1))
    k = mix64(k + 0x8CB92BA72F3D8DD7 * (agent + 1))
    return k


class AgentRandom(_random.Random):
    """Counter-based generator: output i (1-based) = mix64(key + i*GOLDEN)."""

    def __new__(cls, *args, **kwargs):  # _random.Random.__new__ accepts at most one argument
        return super().__new__(cls)

    def __init__(self, seed=0, dom=0, it=0, agent=0):
        self._key = stream_key(int(seed), int(dom), int(it), int(agent))
        self._ctr = 0
        super().__init__(0)

    def seed(self, *args, **kwargs):  # keyed at construction; Random.__init__ calls this
        return None

    def rekey(self, seed, dom, it, agent):
        self._key = stream_key(int(seed), int(dom), int(it), int(agent))
        self._ctr = 0
        return self

    def next64(self):
        self._ctr += 1
        return mix64(self._key + self._ctr * GOLDEN)

    def random(self):
        return (self.next64() >> 11) * (1.0 / 9007199254740992.0)

    def getrandbits(self, k):
        if k <= 0:
            return 0
        if k <= 64:
            return self.next64() >> (64 - k)
        out, have = 0, 0
        while have < k:
            take = min(64, k - have)
            out |= (self.next64() >> (64 - take)) << have
            have += take
        return out

    @property
    def draws(self):
        return self._ctr

    def getstate(self):
        return (self._key, self._ctr)

    def setstate(self, state):
        self._key, self._ctr = state
