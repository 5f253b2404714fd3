"""TEST INFRASTRUCTURE — a deterministic stand-in for the Vicuna sentencepiece tokenizer (absent from the image): words and
punctuation hash to ids in [3, 32000), the eight MMMM special tokens get the ids the reference assigns (tokenizer.py:36-44:
appended after the base vocabulary in the order sys, usr, grd, ngrd, bop, eop, bonp, eonp). Used by oracle/make_golden.py to
drive the reference's prepare_vlm_inputs and by tests/test_inputs_cpu.py to drive the product's."""
import re
import zlib


class FakeTokenizer:
    sys_token, usr_token, grd_token, ngrd_token = '<sys>', '<usr>', '<grd>', '<ngrd>'
    bop_token, eop_token, bonp_token, eonp_token = '<p>', '</p>', '<np>', '</np>'
    bos_token_id, eos_token_id = 1, 2

    def __init__(self, base_vocab_size: int = 32000):
        self.base_vocab_size = base_vocab_size
        names = ['sys', 'usr', 'grd', 'ngrd', 'bop', 'eop', 'bonp', 'eonp']
        self._special = {}
        for i, n in enumerate(names):
            setattr(self, f'{n}_token_id', base_vocab_size + i)
            self._special[getattr(self, f'{n}_token')] = base_vocab_size + i
        self._pat = re.compile('(' + '|'.join(re.escape(t) for t in sorted(self._special, key=len, reverse=True)) + r'|\w+|[^\w\s])')

    def encode(self, text: str, add_special_tokens: bool = False) -> list[int]:
        ids = [self.bos_token_id] if add_special_tokens else []
        for piece in self._pat.findall(text):
            ids.append(self._special[piece] if piece in self._special else 3 + zlib.crc32(piece.encode()) % (self.base_vocab_size - 3))
        return ids
