"""`pytoda.smiles.smiles_language.SMILESLanguage` stand-in: the attributes and
methods the reference uses (examples/train_vae.py:91,131-155,
models/training.py:220-233, examples/generate_latent_codes.py:140-148):
``load`` / ``save``, ``number_of_tokens``, ``padding_index``,
``unknown_index``, ``start_index``, ``stop_index``, ``index_to_token``,
``token_to_index``, ``smiles_to_token_indexes``, ``token_indexes_to_smiles``,
``selfies_to_smiles``.  Token convention of the reference's batch preparation
(utils/utils.py:32-34,184): <PAD>=0, <UNK>=1, <START>=2, <STOP>=3.

Files are JSON (pytoda pickles its language objects; a pickle is never loaded
here)."""
import json
import re

# the usual SMILES token pattern: bracket atoms, two-letter halogens, organic
# subset, aromatic atoms, bonds, branches, ring closures (%NN and digits)
SMILES_TOKENS = re.compile(
    r'(\[[^\]]+\]|Br|Cl|Si|Se|se|@@|%[0-9]{2}|[BCNOPSFIbcnops]|'
    r'[=#$:/\\\-\+\(\)\.~@\?>\*]|[0-9])'
)
SPECIAL = ('<PAD>', '<UNK>', '<START>', '<STOP>')


class SMILESLanguage:

    def __init__(self, name='smiles-language', add_start_and_stop=True):
        self.name = name
        self.add_start_and_stop = add_start_and_stop
        self.padding_token, self.unknown_token = SPECIAL[0], SPECIAL[1]
        self.start_token, self.stop_token = SPECIAL[2], SPECIAL[3]
        self.token_to_index = {tok: i for i, tok in enumerate(SPECIAL)}
        self.index_to_token = {i: tok for i, tok in enumerate(SPECIAL)}
        self.padding_index, self.unknown_index = 0, 1
        self.start_index, self.stop_index = 2, 3
        self.max_token_sequence_length = 0

    # ---- vocabulary -------------------------------------------------------------
    @property
    def number_of_tokens(self):
        return len(self.index_to_token)

    @staticmethod
    def tokenize(smiles):
        tokens = SMILES_TOKENS.findall(smiles)
        if ''.join(tokens) != smiles:
            # characters outside the pattern become single-character tokens
            tokens, pos = [], 0
            for match in SMILES_TOKENS.finditer(smiles):
                tokens.extend(smiles[pos:match.start()])
                tokens.append(match.group(0))
                pos = match.end()
            tokens.extend(smiles[pos:])
        return tokens

    def add_smiles(self, smiles):
        tokens = self.tokenize(smiles)
        for token in tokens:
            if token not in self.token_to_index:
                index = len(self.index_to_token)
                self.token_to_index[token] = index
                self.index_to_token[index] = token
        extra = 2 if self.add_start_and_stop else 0
        self.max_token_sequence_length = max(
            self.max_token_sequence_length, len(tokens) + extra
        )

    def add_smis(self, lines):
        for line in lines:
            self.add_smiles(line)

    # ---- text <-> ids -----------------------------------------------------------
    def smiles_to_token_indexes(self, smiles):
        ids = [
            self.token_to_index.get(token, self.unknown_index)
            for token in self.tokenize(smiles)
        ]
        if self.add_start_and_stop:
            ids = [self.start_index] + ids + [self.stop_index]
        return ids

    def token_indexes_to_smiles(self, token_indexes):
        return ''.join(
            self.index_to_token.get(int(i), '')
            for i in token_indexes if int(i) > self.stop_index
        )

    def selfies_to_smiles(self, selfies):
        """The `selfies` package is not part of this image: text passes
        through unchanged."""
        return selfies

    # ---- persistence ------------------------------------------------------------
    def save(self, filepath):
        with open(filepath, 'w') as fp:
            json.dump({
                'name': self.name,
                'add_start_and_stop': self.add_start_and_stop,
                'max_token_sequence_length': self.max_token_sequence_length,
                'tokens': [self.index_to_token[i]
                           for i in range(self.number_of_tokens)],
            }, fp)

    @classmethod
    def load(cls, filepath):
        with open(filepath, 'rb') as fp:
            head = fp.read(1)
        if head != b'{':
            raise ValueError(
                f'{filepath}: not a JSON language file written by this '
                'stand-in (pytoda pickles are not loaded)'
            )
        with open(filepath) as fp:
            state = json.load(fp)
        language = cls(state['name'], state['add_start_and_stop'])
        if tuple(state['tokens'][:4]) != SPECIAL:
            raise ValueError(f'{filepath}: special tokens must come first')
        for index, token in enumerate(state['tokens']):
            language.token_to_index[token] = index
            language.index_to_token[index] = token
        language.max_token_sequence_length = state['max_token_sequence_length']
        return language
