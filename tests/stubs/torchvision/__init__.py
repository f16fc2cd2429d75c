"""Stand-in for `torchvision`, which is absent from this image.

TEST INFRASTRUCTURE ONLY.  It lets the UNMODIFIED reference driver (`attack.py:29,527`,
`experiments/model.py:22,52-57`, `experiments/dataset.py:25,32-41,113-125`) import and run: the
driver only needs the `models`, `transforms` and `datasets` namespaces, four transform classes and a
dataset class per name it is asked for.  The datasets return seeded synthetic images of the real
datasets' shapes (there is no network and no dataset cache), so a run exercises the whole training
loop — backprop, momentum, attack, aggregation rule, study statistics — on data that is a pure
function of the seed.
"""

import types

import torch

__version__ = "0.0-stub"


class _Transform:
  def __init__(self, *args, **kwargs):
    pass

  def __call__(self, x):
    return x


class _Compose(_Transform):
  def __init__(self, transforms):
    self.transforms = list(transforms)

  def __call__(self, x):
    for t in self.transforms:
      x = t(x)
    return x


class _Normalize(_Transform):
  def __init__(self, mean, std, inplace=False):
    self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

  def __call__(self, x):
    return (x - self.mean) / self.std


transforms = types.SimpleNamespace(
  RandomHorizontalFlip=_Transform,  # identity: keeps a run a pure function of the seed
  ToTensor=_Transform,              # the synthetic images already are fp32 tensors
  Normalize=_Normalize,
  Compose=_Compose)


class _Synthetic(torch.utils.data.Dataset):
  """`count` seeded images in [0, 1] with labels correlated to them (so the loss can move)."""

  shape = (1, 28, 28)
  classes = 10
  count = (2048, 512)  # train, test
  seed = 20210101

  def __init__(self, root, train=True, download=False, transform=None, target_transform=None):
    gen = torch.Generator().manual_seed(self.seed + (0 if train else 1))
    count = self.count[0 if train else 1]
    self.labels = torch.randint(0, self.classes, (count,), generator=gen)
    protos = torch.rand((self.classes,) + self.shape, generator=torch.Generator().manual_seed(self.seed))
    self.images = (0.5 * protos[self.labels] + 0.5 * torch.rand((count,) + self.shape, generator=gen)).contiguous()
    self.transform = transform

  def __len__(self):
    return len(self.labels)

  def __getitem__(self, index):
    image = self.images[index]
    if self.transform is not None:
      image = self.transform(image)
    return image, int(self.labels[index])


class MNIST(_Synthetic):
  pass


class CIFAR10(_Synthetic):
  shape = (3, 32, 32)


class CIFAR100(_Synthetic):
  shape = (3, 32, 32)
  classes = 100


datasets = types.SimpleNamespace(MNIST=MNIST, CIFAR10=CIFAR10, CIFAR100=CIFAR100)
models = types.SimpleNamespace()
