"""The reference's epoch loops as ITS driver runs them, restated for the tests and for ``bench.py --loop reference``.

``north_star`` asks for modules that drop into ``main.py`` / ``trainer.py`` UNCHANGED.  Those files cannot travel to the GPU
box, so this file restates what /root/reference/trainer.py:23-86 (training) and :89-154 (evaluation) do to a model, step by
step, in the build's own words — nothing here knows about gnan_amd:

* the whole training epoch runs inside ``torch.autograd.set_detect_anomaly(True)`` (trainer.py:24);
* per batch: the label rules (:32-40), ``data.to(device)`` (:46), ``optimizer.zero_grad()`` (:48), ``model.forward(data)``
  (:52), the task mask applied AFTER the forward for node tasks (:53-55), the flatten rule of the loss (:61-64),
  ``loss.backward()``, ``optimizer.step()`` (:66-67) with whatever stock optimizer the caller built over
  ``model.parameters()`` (main.py:141), one ``loss.item()`` per batch (:72) and the hit count (:5-20, :74-75);
* evaluation: ``torch.no_grad()``, ``model.eval()`` and never ``train()`` again (:90, :97), validation or test mask (:125-131).

Test infrastructure (like ``oracle/``): the product never imports it.
"""
import torch


def _labels(data, label_index, loss_fn):
    y = data.y
    labels = y[:, label_index].reshape(-1).float() if y.dim() > 1 else y.reshape(-1)
    if bool((labels == -1).any()):
        labels = (labels + 1) / 2
    return labels.long() if type(loss_fn).__name__ == "CrossEntropyLoss" else labels


def _hit_count(outputs, labels):
    if outputs.dim() == 2 and outputs.shape[-1] > 1:
        return int((torch.softmax(outputs, dim=-1).argmax(dim=-1) == labels).sum())
    return int(((torch.sigmoid(outputs).reshape(-1) > 0.5) == labels).sum())


def _loss(loss_fn, outputs, labels):
    if outputs.dim() == 2 and outputs.shape[-1] == 1:
        return loss_fn(outputs.flatten(), labels.float())
    return loss_fn(outputs, labels)


def train_epoch(model, dloader, loss_fn, optimizer, device, classify=True, label_index=0, is_graph_task=True,
                detect_anomaly=True):
    """-> (mean loss per batch, hits / samples, -1) for classification, (mean loss, -1) otherwise (trainer.py:80-86, no AUC)."""
    loss_sum, hits, samples = 0.0, 0, 0
    with torch.autograd.set_detect_anomaly(detect_anomaly):
        for data in dloader:
            labels = _labels(data, label_index, loss_fn)
            data, labels = data.to(device), labels.to(device)
            optimizer.zero_grad()
            outputs = model.forward(data)
            if not is_graph_task:
                labels, outputs = labels[data.train_mask], outputs[data.train_mask]
            samples += len(labels)
            loss = _loss(loss_fn, outputs, labels)
            loss.backward()
            optimizer.step()
            loss_sum += loss.item()
            if classify:
                hits += _hit_count(outputs, labels)
    return (loss_sum / len(dloader), hits / samples, -1) if classify else (loss_sum / len(dloader), -1)


def test_epoch(model, dloader, loss_fn, device, classify=True, label_index=0, val_mask=False, is_graph_task=True):
    loss_sum, hits, samples = 0.0, 0, 0
    with torch.no_grad():
        model.eval()
        for data in dloader:
            labels = _labels(data, label_index, loss_fn)
            data, labels = data.to(device), labels.to(device)
            outputs = model.forward(data)
            if not is_graph_task:
                mask = data.val_mask if val_mask else data.test_mask
                labels, outputs = labels[mask], outputs[mask]
            samples += len(labels)
            loss_sum += _loss(loss_fn, outputs, labels).item()
            if classify:
                hits += _hit_count(outputs, labels)
    return (loss_sum / len(dloader), hits / samples, -1) if classify else (loss_sum / len(dloader), -1)


test_epoch.__test__ = False          # (not a pytest case)
